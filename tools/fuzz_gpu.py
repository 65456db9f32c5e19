"""GPU fuzz: random scenes with awkward scales / offsets / duplicates against the oracle's
exhaustive walk, every bounds mode, nearest + kNN + a small render.  Prints the first mismatch."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S
from oracle import oracle as O

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)

def run(seed, iters, gpu=None, log=print, heavy=False):
  """heavy: larger trees now and then, a render on every other scene with the camera sometimes inside the
  scene, t_min zero or not, and a varying step cap (the soak run, tools/soak_gpu.sh)."""
  rng = np.random.default_rng(seed)
  gpu = gpu or S.SdfGpu(0)
  bad = 0
  last_pos = None
  for it in range(iters):
      n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 1000] + ([4000] if heavy else [])))
      scale = float(rng.choice([1e-3, 1.0, 50.0, 1e4]))
      offset = rng.choice([0.0, 1e3, -1e5]) * rng.uniform(-1, 1, 3)
      # heavy: a third of the scenes as the crate's own builder emits them (PCA trans_inverse in every merged box):
      # the chain path where every map is an isometry, the exhaustive general path otherwise
      pca = heavy and rng.random() < 0.33
      popts = {"pca_trans_inverse": True} if pca else None
      # heavy: a fifth of the other scenes with the opt-in extensions in effect (smooth unions with a radius from far
      # below to far above the primitives' spacing, box frames, one lattice clone): the general path, whose walks
      # prune from 256 resident nodes on (kernels.hip walk_eval_pruned / walk_general_pruned)
      ext = 0
      smooth_k = 0.0
      if heavy and not pca and rng.random() < 0.2:
          ext = int(rng.choice([S.EXT_SMOOTH_UNION, S.EXT_BOX_FRAME, S.EXT_CLONE_REPAIR, S.EXT_SMOOTH_UNION | S.EXT_BOX_FRAME | S.EXT_CLONE_REPAIR]))
          if ext & S.EXT_SMOOTH_UNION:
              smooth_k = scale * float(rng.choice([0.02, 0.5, 2.0, 30.0]))
      parts = []
      for k in range(n):
          size = scale * float(rng.choice([1e-3, 0.05, 1.0])) * rng.uniform(0.5, 2.0)
          e = S.SdfSphere(size) if rng.random() < 0.5 else S.SdfBox(tuple(float(v) for v in size * rng.uniform(0.2, 2.0, 3)))
          if (ext & S.EXT_BOX_FRAME) and (k == 0 or rng.random() < 0.25):
              e = S.SdfBoxFrame(tuple(float(v) for v in size * rng.uniform(0.5, 2.0, 3)), float(0.1 * size))
          pos = offset + scale * 20 * rng.uniform(-1, 1, 3) if (k == 0 or rng.random() > 0.15) else last_pos   # 15% exact duplicates of a position
          last_pos = pos
          q = S.Quat.from_euler_xyz(*(float(v) for v in rng.uniform(0, 6.28, 3)))
          t = S.Transform.from_translation(tuple(float(v) for v in pos))
          t.rotation = q
          leaf = S.SdfBuilder.primitive(e, opts=popts).transform(t)
          if (ext & S.EXT_CLONE_REPAIR) and k == n // 2:
              d = tuple(float(v) for v in scale * rng.uniform(2.0, 8.0, 3))
              leaf = S.SdfBuilder.primitive(e).operation(S.SdfCaaClone(d, (-1, -2, 0), (2, 1, 1)))
          parts.append(leaf)
      b = parts[0]
      if n > 1:
          b = b.operation(S.SdfUnion(smooth_k))
          for x in parts[1:]:
              b = b.with_(x)
      buf = b.finalize().expanded().make_buffer()
      m = 3000
      pts = np.ones((m, 4), np.float32)
      spread = float(rng.choice([1.0, 30.0, 1e3]))
      pts[:, :3] = (offset + scale * spread * rng.uniform(-1, 1, (m, 3))).astype(np.float32)
      if heavy:   # points far outside the scene: the unclamped copy of the walk (kernels.hip, FAR)
          u = rng.normal(size=(150, 3))
          u /= np.linalg.norm(u, axis=1, keepdims=True)
          pts[-150:, :3] = (offset + u * scale * 20 * rng.choice([20.0, 300.0, 3e4], size=(150, 1))).astype(np.float32)
      pts[: min(n, 50), :3] = [buf.downtree_buffer[i]["bounding_box"]["matrix"][12:15] for i in np.flatnonzero(buf.downtree_buffer["is_primitive"])[: min(n, 50)]]  # primitive centres
      k = int(rng.choice([1, 3, 8, 16]))
      with O.extensions(ext):
          d0, i0 = O.nearest(buf, pts, mode=O.MODE_EXHAUSTIVE, threads=8)
          dk0, ik0 = O.knn(buf, pts[:800], k, threads=8)
      for mode in ((S.BOUNDS_AUTO,) if ext else (S.BOUNDS_AUTO, S.BOUNDS_NONE) if pca else (S.BOUNDS_BVH, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE)):
          gpu.upload_tree(buf, bounds=mode, extensions=ext)
          d, i = gpu.nearest(pts)
          if not (np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)):
              j = int(np.flatnonzero((bits(d) != bits(d0)) | (i != i0))[0])
              log(f"MISMATCH nearest it={it} n={n} scale={scale} mode={mode} pt={pts[j]} gpu=({d[j]},{i[j]}) oracle=({d0[j]},{i0[j]})")
              bad += 1
          dk, ik = gpu.knn(pts[:800], k)
          if not (np.array_equal(bits(dk), bits(dk0)) and np.array_equal(ik, ik0)):
              log(f"MISMATCH knn it={it} n={n} scale={scale} mode={mode} k={k}")
              bad += 1
      if it % (2 if heavy else 6) == 0:
          w, h = 72, 40
          L = scale * 20
          eye = np.array([0.3 * L, 0.2 * L, -3.0 * L])
          steps, t_min = 64, 0.0
          if heavy:
              if rng.random() < 0.4:
                  eye = L * rng.uniform(-0.8, 0.8, 3)          # inside the scene, perhaps inside a primitive
              steps = int(rng.choice([16, 64, 200]))
              t_min = float(rng.choice([0.0, 0.0, 0.5 * L]))
          cam = S.camera_look_at(tuple(float(v) for v in offset + eye), tuple(float(v) for v in offset), (0, 1, 0), 0.8, w, h)
          mp = S.march_params(max_steps=steps, eps_hit=1e-3 * scale, t_min=t_min, t_max=8 * L, normal_h=1e-3 * scale)
          with O.extensions(ext):
              ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
          for mode in ((S.BOUNDS_AUTO,) if (pca or ext) else (S.BOUNDS_BVH, S.BOUNDS_OBB)):
              gpu.upload_tree(buf, bounds=mode, extensions=ext)
              img = gpu.render(cam, w, h, mp)
              ok = all(np.array_equal(img[q], ref[q]) for q in ("hit", "prim")) and np.array_equal(bits(img["depth"]), bits(ref["depth"])) and np.array_equal(bits(img["normal"]), bits(ref["normal"]))
              if not ok:
                  log(f"MISMATCH render it={it} n={n} scale={scale} mode={mode} hits={int(ref['hit'].sum())}")
                  bad += 1
  return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60,
                heavy=len(sys.argv) > 3 and sys.argv[3] == "heavy")
    print("fuzz done, mismatches:", n_bad)
    sys.exit(1 if n_bad else 0)
