"""Second, independent CPU restatement of the flat query -- TEST INFRASTRUCTURE ONLY.

Written directly from the reference source (sdf/src/faux_shader.rs), not from
oracle/oracle.cpp, in numpy float32 with ONE rounded operation per line, so that
a misreading in one restatement shows up as a bit difference against the other
(tests/test_second_restatement.py).  It does NOT pin parity: the reference holds
no golden vectors, fixtures or seeded tests for this path and cannot be built in
this image, so parity stays "unpinned"; this file only removes
the single-restatement risk.

Only tests/ may import it.  Arithmetic orders of glam 0.20 (pulled in by
bevy 0.6, sdf/Cargo.toml:11; SSE2 build) as this file assumes them:
  Mat4 * Vec4       = ((c0*x + c1*y) + c2*z) + c3*w          per component
  Vec4::length      = sqrt((x*x + z*z) + (y*y + w*w))         (shuffle-add dot)
  Vec3::length      = sqrt((x*x + y*y) + z*z)                 (scalar Vec3)
"""
import numpy as np

F = np.float32
INF = F(np.inf)
NO_PRIM = 0xFFFFFFFF

# Opt-in, build-defined extensions (include/sdfgpu.h sdfgpu_extension); 0 = the crate as written.
EXT_SMOOTH_UNION, EXT_BOX_FRAME, EXT_CLONE_REPAIR = 1, 2, 4
EXT = 0


def set_extensions(mask):
    global EXT
    old, EXT = EXT, int(mask)
    return old


def _mat4_mul_vec4(m, v):
    """glam Mat4 * Vec4 (faux_shader.rs:50,55,165,175).  m: 16 floats column-major; v: (..., 4)."""
    c0, c1, c2, c3 = m[0:4], m[4:8], m[8:12], m[12:16]
    x, y, z, w = (v[..., k:k + 1] for k in range(4))
    r = c0 * x              # one rounded product per component
    t = c1 * y
    r = r + t
    t = c2 * z
    r = r + t
    t = c3 * w
    r = r + t
    return r.astype(F)


def _length4(v):
    """glam SSE2 Vec4::length: dot via shuffle-adds, then sqrt."""
    s = v * v
    a = s[..., 0] + s[..., 2]
    b = s[..., 1] + s[..., 3]
    return np.sqrt(a + b).astype(F)


def _length3(x, y, z):
    """glam scalar Vec3::length (point.truncate().length(), faux_shader.rs:63)."""
    a = x * x
    b = y * y
    a = a + b
    b = z * z
    return np.sqrt(a + b).astype(F)


def mindist(bbox, pts):
    """faux_shader.rs:54-58.  bbox: one record of BBOX_DTYPE; pts: (n, 4) float32."""
    trans = _mat4_mul_vec4(bbox["full_inverse"].astype(F), pts)
    q = np.abs(trans)
    q = q - F(1.0)
    q = q * bbox["scale"].astype(F)
    l = _length4(np.where(q > 0, q, F(0.0)).astype(F))          # q_local.max(Vec4::ZERO).length()
    m = np.maximum(q[..., 1], q[..., 2])                         # q.y.max(q.z)
    m = np.maximum(q[..., 0], m)                                 # q.x.max(..)
    m = np.minimum(m, F(0.0))                                    # .min(0.0)
    return (l + m).astype(F)


def maxdist(bbox, pts):
    """faux_shader.rs:39-52: farthest of the eight corners, in the listed order, CmpFloat max."""
    best = None
    for sz in (1.0, -1.0):
        for sy in (1.0, -1.0):
            for sx in (1.0, -1.0):
                corner = _mat4_mul_vec4(bbox["matrix"].astype(F), np.array([[sx, sy, sz, 1.0]], F))
                d = _length4((corner - pts).astype(F))
                best = d if best is None else np.where(d > best, d, best)
    return best.astype(F)


def prim_value(op_code, op_specific, p):
    """faux_shader.rs:60-67 for op 0; op 2 = the build-defined solid box, which reuses the crate's
    only box arithmetic (obb.rs:297-299) on xyz with the half extents in vec4s[0]."""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    if op_code == 0:
        return (_length3(x, y, z) - F(op_specific["floats"][0])).astype(F)
    if op_code == 2:
        h = op_specific["vec4s"][0].astype(F)
        qx = np.abs(x) - h[0]
        qy = np.abs(y) - h[1]
        qz = np.abs(z) - h[2]
        zero = F(0.0)
        l = _length3(np.where(qx > 0, qx, zero).astype(F), np.where(qy > 0, qy, zero).astype(F),
                     np.where(qz > 0, qz, zero).astype(F))
        m = np.maximum(qy, qz)
        m = np.maximum(qx, m)
        m = np.minimum(m, zero)
        return (l + m).astype(F)
    if op_code == 5 and (EXT & EXT_BOX_FRAME):
        # box frame of (dimension b, thickness e): p = |p| - b, q = |p + e| - e, nearest of the three bar families
        b = op_specific["vec4s"][0].astype(F)
        e = F(op_specific["floats"][0])
        px = np.abs(x) - b[0]
        py = np.abs(y) - b[1]
        pz = np.abs(z) - b[2]
        qx = np.abs(px + e) - e
        qy = np.abs(py + e) - e
        qz = np.abs(pz + e) - e
        zero = F(0.0)

        def bar(a, bb, c):
            l = _length3(np.where(a > 0, a, zero).astype(F), np.where(bb > 0, bb, zero).astype(F),
                         np.where(c > 0, c, zero).astype(F))
            m = np.maximum(bb, c)
            m = np.maximum(a, m)
            m = np.minimum(m, zero)
            return (l + m).astype(F)

        d0, d1, d2 = bar(px, qy, qz), bar(qx, py, qz), bar(qx, qy, pz)
        d = np.where(d0 < d1, d0, d1)
        return np.where(d < d2, d, d2).astype(F)
    raise ValueError(f"Unsupported primitive op code: {op_code}")


def leaf_chain(buf, pts):
    """Every primitive's value at every point: the point handed down level by level, each node applying
    its own trans_inverse (faux_shader.rs:165,175; union downtree = [p, p], :72).  Returns (n_prims, n)
    float32 in pre-order primitive order.  Clone subtrees are left out (+inf as written, :90)."""
    dt = buf.downtree_buffer
    pts = np.ascontiguousarray(pts, F).reshape(-1, 4)
    frames = {1: pts}
    out = []
    dead_below = None     # level of a clone node whose subtree is being skipped
    for i in range(buf.buffer_len):
        b = dt[i]
        lv = int(b["level"])
        if dead_below is not None and lv > dead_below:
            if b["is_primitive"]:
                out.append(np.full(len(pts), INF, F))
            continue
        dead_below = None
        p = _mat4_mul_vec4(b["bounding_box"]["trans_inverse"].astype(F), frames[lv])
        if b["is_primitive"]:
            out.append(prim_value(int(b["op_code"]), b["op_specific"], p))
        elif int(b["op_code"]) == 1 and not (EXT & EXT_CLONE_REPAIR):
            dead_below = lv          # uptree op 1 returns the (never filled) right branch: +inf
        elif int(b["op_code"]) == 1:
            v = b["op_specific"]["vec4s"].astype(F)
            with np.errstate(all="ignore"):
                r = np.rint(p / v[0]).astype(F)
                r = np.minimum(np.maximum(r, v[1]), v[2]).astype(F)
                frames[lv + 1] = (p - v[0] * r).astype(F)
        else:
            frames[lv + 1] = p
    return np.array(out, F).reshape(len(out), len(pts))


def nearest_exhaustive(buf, pts):
    """min over primitives with the reference's combine composed over the tree: `if a < b {a} else {b}`
    (faux_shader.rs:27-29) = the LAST primitive in pre-order attaining the minimum."""
    v = leaf_chain(buf, pts)
    n = np.asarray(pts).reshape(-1, 4).shape[0]
    best = np.full(n, INF, F)
    idx = np.full(n, NO_PRIM, np.uint32)
    for j in range(v.shape[0]):
        keep_old = best < v[j]          # a < b ? a : b  with a = what came before
        idx = np.where(keep_old, idx, np.uint32(j)).astype(np.uint32)
        best = np.where(keep_old, best, v[j]).astype(F)
    return best, idx


def faux_nearest_scalar(buf, point, prune=True):
    """faux_shader::nearest_neighbor (faux_shader.rs:96-208) as written, for ONE point: the literal state
    machine with its 256 frames, prune test and up-pass.  Pure-Python loop: small cases only."""
    dt, ut, n = buf.downtree_buffer, buf.uptree_buffer, buf.buffer_len
    pts = [[np.zeros(4, F), np.zeros(4, F)] for _ in range(256)]
    dists = [[F(0.0), F(0.0)] for _ in range(256)]
    fill = [0] * 256
    pts[1] = [np.asarray(point, F).reshape(4), np.zeros(4, F)]
    dists[1] = [INF, INF]
    fill[1] = 0

    def up(code, spec, l, r):
        if code == 0:
            m = l if l < r else r
            k = F(spec["floats"][0])
            if (EXT & EXT_SMOOTH_UNION) and k > 0:
                with np.errstate(all="ignore"):
                    t = k - np.abs(F(l - r))
                    t = t if t > 0 else F(0.0)
                    h = F(t / k)
                    q = F(h * h)
                    q = F(q * k)
                    q = F(q * F(0.25))
                    m = F(m - q)
            return m
        if code == 1:
            return l if (EXT & EXT_CLONE_REPAIR) else r
        raise ValueError("Unsupported downtree op code")

    di = ui = last_level = 0
    while di < n:
        b = dt[di]
        lv = int(b["level"])
        p = pts[lv][fill[lv]]
        cmp_ = dists[lv][0]
        if prune and b["parent_is_union"]:      # prune=False: the exhaustive walk (the contract)
            m = mindist(b["bounding_box"], p.reshape(1, 4))[0]
            if m > 0 and (m > cmp_ or m > maxdist(b["other_box"], p.reshape(1, 4))[0]):
                di += 1 + int(b["len"])
                ui += 1 + int(b["len"])
                continue
        if lv < last_level:
            last_ut = 1 << 32
            while True:
                u = ut[ui]
                ul = int(u["level"])
                if ul >= last_ut:
                    break
                l, r = dists[ul + 1]
                dists[ul][fill[ul]] = up(int(u["op_code"]), u["op_specific"], l, r)
                fill[ul] += 1
                last_ut = ul
                ui += 1
        q = _mat4_mul_vec4(b["bounding_box"]["trans_inverse"].astype(F), p.reshape(1, 4))[0]
        if b["is_primitive"]:
            dists[lv][fill[lv]] = prim_value(int(b["op_code"]), b["op_specific"], q.reshape(1, 4))[0]
            fill[lv] += 1
            ui += 1
        else:
            code = int(b["op_code"])
            if code == 0:
                pts[lv + 1] = [q, q]
            elif code == 1:
                v = b["op_specific"]["vec4s"].astype(F)
                with np.errstate(all="ignore"):
                    r = np.rint(q / v[0]).astype(F)          # glam SSE2 round: half to even
                    r = np.maximum(v[1], np.minimum(r, v[2])).astype(F)
                    pts[lv + 1] = [(q - v[0] * r).astype(F), np.zeros(4, F)]
            else:
                raise ValueError("Unsupported downtree op code")
            fill[lv + 1] = 0
            dists[lv + 1] = [INF, INF]
        last_level = lv
        di += 1
    while ui < n:
        u = ut[ui]
        ul = int(u["level"])
        l, r = dists[ul + 1]
        dists[ul][fill[ul]] = up(int(u["op_code"]), u["op_specific"], l, r)
        fill[ul] += 1
        ui += 1
    return dists[1][0]
