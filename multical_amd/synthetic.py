"""Synthetic multi-camera rigs for benchmarks, tests and golden-vector generation.

The reference ships no sample data; inputs of the shape BASELINE.json names are
synthesised here following SURVEY.md section 8(d): seeded `np.random.default_rng(seed)`, 2000x1500 images,
K ~ [[2250,0,1000],[0,2250,750]] jittered per camera, Brown-Conrady / Kannala-Brandt distortion, 0.2 px
Gaussian noise, per-view Bernoulli(0.7) and per-point Bernoulli(0.9) visibility, 1 % gross outliers, and
an initial guess = truth perturbed by N(0, 0.01 rad / 5 mm) on every pose and +-0.5 % on intrinsics.

Pose convention (reference: multical/optimization/calibration.py:87-90, motion/static_frames.py:16-25):
    X_cam = camera_pose[c] @ rig_pose[f] @ board_pose[b] @ [X_board; 1]

The small numpy projector in this file exists only to *synthesise observations*; it is not used by the
solver (multical_amd.backend -> HIP) nor by the oracle (oracle/restate.py has its own restatement).
"""
from types import SimpleNamespace
import numpy as np
from scipy.spatial.transform import Rotation as R

IMAGE_SIZE = (2000, 1500)


# ------------------------------------------------------------------------------------------------
# board geometry (reference: board/charuco.py:56-58 -> cv2 chessboardCorners, float32;
#                 board/aprilgrid.py:78-83 + aprilgrid_detector.py:44-55, float64)
# ------------------------------------------------------------------------------------------------
def charuco_points(size, square_length):
  w, h = size
  pts = [((i + 1) * square_length, (j + 1) * square_length, 0.0) for j in range(h - 1) for i in range(w - 1)]
  return np.array(pts, dtype=np.float32)  # OpenCV stores chessboardCorners as Point3f


def aprilgrid_points(size, tag_length, tag_spacing):
  w, h = size
  columns = w
  a = tag_length
  b = tag_spacing * a
  corners = []
  for tag_id in range(w * h):
    row, col = tag_id // columns, tag_id % columns
    lo = lambda i: i * (a + b)
    hi = lambda i: (i + 1) * a + i * b
    corners += [(lo(col), lo(row)), (hi(col), lo(row)), (hi(col), hi(row)), (lo(col), hi(row))]
  p2 = np.array(corners, dtype=np.float64).reshape(-1, 2)
  return np.concatenate([p2, np.zeros((p2.shape[0], 1))], axis=1)


BOARDS = dict(
  charuco_16x22=lambda: charuco_points((16, 22), 0.025),       # example_boards/charuco_16x22.yaml
  aprilgrid_9x9=lambda: aprilgrid_points((9, 9), 0.06, 0.3),   # example_boards/aprilgrid_9x9.yaml
  charuco_10x10=lambda: charuco_points((10, 10), 0.040),       # example_boards/cube_10x10.yaml (x5)
  charuco_25x35=lambda: charuco_points((25, 35), 0.020),       # 24 x 34 = 816 corners: more than one 512-slot segment
)


# ------------------------------------------------------------------------------------------------
# SE(3) helpers
# ------------------------------------------------------------------------------------------------
def to_matrix(rtvec):
  rtvec = np.asarray(rtvec, dtype=np.float64)
  m = np.zeros(rtvec.shape[:-1] + (4, 4))
  m[..., :3, :3] = R.from_rotvec(rtvec[..., :3].reshape(-1, 3)).as_matrix().reshape(rtvec.shape[:-1] + (3, 3))
  m[..., :3, 3] = rtvec[..., 3:]
  m[..., 3, 3] = 1.0
  return m


def perturb(poses, rng, rot_sigma, trans_sigma):
  d = np.concatenate([rng.normal(0, rot_sigma, poses.shape[:-2] + (3,)),
                      rng.normal(0, trans_sigma, poses.shape[:-2] + (3,))], axis=-1)
  return to_matrix(d) @ poses


def look_rotation(forward, up=(0.0, -1.0, 0.0)):
  """Rotation (world<-camera columns) whose +z axis points along `forward`."""
  z = np.asarray(forward, dtype=np.float64)
  z = z / np.linalg.norm(z)
  x = np.cross(np.asarray(up, dtype=np.float64), z)
  x = x / np.linalg.norm(x)
  y = np.cross(z, x)
  return np.stack([x, y, z], axis=1)


# ------------------------------------------------------------------------------------------------
# projection used to synthesise observations
# ------------------------------------------------------------------------------------------------
def project_pinhole(X, K, dist):
  k = np.zeros(14)
  k[:len(dist)] = dist
  x, y = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
  r2 = x * x + y * y
  r4, r6 = r2 * r2, r2 * r2 * r2
  radial = (1 + k[0] * r2 + k[1] * r4 + k[4] * r6) / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6)
  xd = x * radial + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r4
  yd = y * radial + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r4
  return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=-1)


def project_fisheye(X, K, dist):
  a, b = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
  r = np.sqrt(a * a + b * b)
  th = np.arctan(r)
  th2 = th * th
  thd = th * (1 + dist[0] * th2 + dist[1] * th2**2 + dist[2] * th2**3 + dist[3] * th2**4)
  s = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
  return np.stack([K[0, 0] * a * s + K[0, 2], K[1, 1] * b * s + K[1, 2]], axis=-1)


def _project(cam, X):
  f = project_fisheye if cam.model == 'fisheye' else project_pinhole
  with np.errstate(all='ignore'):
    return f(X, cam.intrinsic, cam.dist)


# ------------------------------------------------------------------------------------------------
# configs (BASELINE.json `configs`)
# ------------------------------------------------------------------------------------------------
CONFIGS = {
  # name: cameras, frames, boards, motion, camera model, optimise intrinsics, layout
  "cfg1": dict(cameras=2, frames=20, boards=["charuco_16x22"], motion="static", model="standard",
               optimize_cameras=False, layout="stereo", seed=1),
  "cfg2": dict(cameras=4, frames=200, boards=["charuco_16x22"], motion="static", model="standard",
               optimize_cameras=True, layout="stereo", seed=2),
  "cfg3": dict(cameras=8, frames=500, boards=["charuco_16x22", "aprilgrid_9x9"], motion="rolling",
               model="standard", optimize_cameras=True, layout="stereo", seed=3),
  "cfg4": dict(cameras=16, frames=1000, boards=["charuco_10x10"] * 5, motion="static", model="standard",
               optimize_cameras=True, layout="stereo", cube=True, seed=4),
  # "outlier pose rejection": views whose PnP pose reprojects worse than pose_error_limit never reach the initialisation
  # tables (tables.py:44-56; the reference's default limit, config/arguments.py:50)
  "cfg5": dict(cameras=6, frames=400, boards=["charuco_10x10"] * 5, motion="static", model="fisheye",
               optimize_cameras=True, layout="ring", seed=5, pose_error_limit=1.0, pose_noise=(2e-4, 1e-4)),
  # small variants used by unit tests / smoke (same generators, fewer frames)
  "tiny": dict(cameras=2, frames=6, boards=["charuco_10x10"], motion="static", model="standard",
               optimize_cameras=True, layout="stereo", seed=11),
  "tiny_rolling": dict(cameras=3, frames=8, boards=["charuco_10x10", "aprilgrid_9x9"], motion="rolling",
                       model="standard", optimize_cameras=True, layout="stereo", seed=12),
  "tiny_fisheye": dict(cameras=3, frames=10, boards=["charuco_10x10"] * 2, motion="static", model="fisheye",
                       optimize_cameras=True, layout="ring", seed=13),
  "tiny_handeye": dict(cameras=2, frames=12, boards=["charuco_10x10"], motion="hand_eye", model="standard",
                       optimize_cameras=False, layout="stereo", seed=14),
  "tiny_rational": dict(cameras=2, frames=8, boards=["charuco_10x10"], motion="static", model="rational",
                        optimize_cameras=True, layout="stereo", seed=15),
  "tiny_tilted": dict(cameras=2, frames=8, boards=["charuco_10x10"], motion="static", model="tilted",
                      optimize_cameras=True, layout="stereo", seed=16),
  # four distortion coefficients (k1 k2 p1 p2): a `standard` reference Camera whose dist array has 4 entries, as loaded
  # from a calibration file written by OpenCV with CALIB_FIX_K3 (cv2.projectPoints accepts 4, 5, 8, 12 or 14)
  "tiny_pin4": dict(cameras=2, frames=8, boards=["charuco_10x10"], motion="static", model="pin4",
                    optimize_cameras=True, layout="stereo", seed=18),
  # rigs beyond the former limits of the HIP back-end (the reference has none): a board with more than 512 points
  # (tables.stack_boards pads to the largest board, tables.py:385-394) next to a small one, and more than 128
  # (camera, board) pairs
  "tiny_bigboard": dict(cameras=2, frames=6, boards=["charuco_25x35", "charuco_10x10"], motion="rolling",
                        model="standard", optimize_cameras=True, layout="stereo", seed=31),
  "tiny_manypairs": dict(cameras=16, frames=10, boards=["charuco_10x10"] * 10, motion="static", model="standard",
                         optimize_cameras=True, layout="stereo", board_grid=5, distance=3.0, seed=32),
  # cameras of DIFFERENT models in one rig (a ParamList of independent Camera objects, optimization/parameters.py:54-85):
  # 5, 8, 14 and 4 distortion coefficients -- the cameras block of the parameter vector is ragged
  "tiny_mixed": dict(cameras=4, frames=8, boards=["charuco_10x10"], motion="static",
                     model=["standard", "rational", "tilted", "pin4"], optimize_cameras=True, layout="stereo", seed=33),
  # pinhole AND fisheye cameras in one rig (independent Camera / CameraFisheye objects in the reference's ParamList): 5 and 4
  # distortion coefficients, two projection families; rolling shutter so that both chains are exercised
  # (4-coefficient pinhole + fisheye: every camera block has 9 entries, the one mixed rig the reference's own bundle_adjust
  #  can solve -- Calibration.sparsity_matrix reshapes the cameras block to [n_cameras, -1], calibration.py:179-180)
  "tiny_fishmix": dict(cameras=4, frames=10, boards=["charuco_10x10"] * 2, motion="rolling",
                       model=["pin4", "fisheye", "pin4", "fisheye"], optimize_cameras=True, layout="stereo", seed=34),
  # (5- and 8-coefficient pinhole + fisheye: ragged cameras block -- the reference evaluates it, its bundle_adjust raises)
  "tiny_fishmix5": dict(cameras=4, frames=8, boards=["charuco_10x10"], motion="static",
                        model=["standard", "fisheye", "rational", "fisheye"], optimize_cameras=True, layout="stereo", seed=35),
  # reduced-frame variants of BASELINE configs[2..4] whose complete reference run (adjust_outliers) finishes in minutes
  "cfg3_40": dict(cameras=8, frames=40, boards=["charuco_16x22", "aprilgrid_9x9"], motion="rolling",
                  model="standard", optimize_cameras=True, layout="stereo", seed=3),
  "cfg4_40": dict(cameras=16, frames=40, boards=["charuco_10x10"] * 5, motion="static", model="standard",
                  optimize_cameras=True, layout="stereo", cube=True, seed=4),
  "cfg5_40": dict(cameras=6, frames=40, boards=["charuco_10x10"] * 5, motion="static", model="fisheye",
                  optimize_cameras=True, layout="ring", seed=5, pose_error_limit=1.0, pose_noise=(2e-4, 1e-4)),
}

DIST_SIZE = dict(standard=5, rational=8, thin_prism=12, tilted=14, fisheye=4, pin4=4)


def _make_camera(model, rng, focal=2250.0):
  w, h = IMAGE_SIZE
  jit = lambda: 1.0 + rng.uniform(-0.01, 0.01)
  K = np.array([[focal * jit(), 0.0, w / 2 * jit()], [0.0, focal * jit(), h / 2 * jit()], [0.0, 0.0, 1.0]])
  if model == 'fisheye':
    dist = np.array([0.05, 0.01, -0.005, 0.001]) * rng.uniform(0.8, 1.2, 4)
  else:
    dist = np.zeros(DIST_SIZE[model])
    n5 = min(5, dist.size)
    dist[:n5] = (np.array([-0.12, 0.3, 1e-3, -1e-3, -0.2]) * rng.uniform(0.8, 1.2, 5))[:n5]
    if model in ('rational', 'thin_prism', 'tilted'):
      dist[5:8] = np.array([0.02, -0.01, 0.005]) * rng.uniform(0.8, 1.2, 3)
    if model in ('thin_prism', 'tilted'):
      dist[8:12] = np.array([1e-3, -5e-4, 8e-4, 3e-4]) * rng.uniform(0.8, 1.2, 4)
    if model == 'tilted':
      dist[12:14] = np.array([0.01, -0.008]) * rng.uniform(0.8, 1.2, 2)
  # (the reference's Camera.model enumerates OpenCV calibration flags; a 4-coefficient camera is a `standard` one)
  return SimpleNamespace(model='standard' if model == 'pin4' else model, image_size=IMAGE_SIZE, intrinsic=K, dist=dist,
                         fix_aspect=False, has_skew=False)


def _perturb_camera(cam, rng):
  K = cam.intrinsic.copy()
  for (i, j) in [(0, 0), (1, 1), (0, 2), (1, 2)]:
    K[i, j] *= 1.0 + rng.uniform(-0.005, 0.005)
  dist = cam.dist * (1.0 + rng.uniform(-0.005, 0.005, cam.dist.shape))
  return SimpleNamespace(model=cam.model, image_size=cam.image_size, intrinsic=K, dist=dist,
                         fix_aspect=cam.fix_aspect, has_skew=cam.has_skew)


def _cube_board_poses(n, side):
  """Faces of a cube of the given side; pose maps board coords (board plane z=0, origin at a corner)."""
  h = side / 2
  faces = [  # (normal, in-plane x axis)
    ((0, 0, -1), (1, 0, 0)), ((-1, 0, 0), (0, 0, -1)), ((1, 0, 0), (0, 0, 1)),
    ((0, -1, 0), (1, 0, 0)), ((0, 1, 0), (1, 0, 0))][:n]
  poses = []
  for nrm, ax in faces:
    nrm, ax = np.array(nrm, float), np.array(ax, float)
    ay = np.cross(-nrm, ax)  # board z axis = -normal (board faces along -z like a flat board seen from -z)
    Rm = np.stack([ax, ay, -nrm], axis=1)
    centre = nrm * h
    origin = centre - Rm @ np.array([h, h, 0.0])
    m = np.eye(4)
    m[:3, :3] = Rm
    m[:3, 3] = origin + np.array([h, h, h])  # shift the cube so that face 0 starts at the origin plane
    poses.append(m)
  return np.stack(poses)


def make_rig(name_or_cfg, frames=None, seed=None, noise=0.2, outlier_frac=0.01, obs_frames=None):
  """Returns SimpleNamespace(truth=..., init=..., points=..., valid=..., ...) of plain numpy data.

  obs_frames=(f0, f1): synthesise observations only for that frame range (poses / parameters cover all frames)."""
  cfg = dict(CONFIGS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)
  if frames is not None:
    cfg["frames"] = frames
  if seed is not None:
    cfg["seed"] = seed
  rng = np.random.default_rng(cfg["seed"])
  C, F = cfg["cameras"], cfg["frames"]
  board_pts = [BOARDS[b]() for b in cfg["boards"]]
  B = len(board_pts)
  P = max(p.shape[0] for p in board_pts)
  ring = cfg.get("layout") == "ring"

  # `model` may be a list: one model per camera (the reference's cameras are independent objects)
  models = cfg["model"] if isinstance(cfg["model"], (list, tuple)) else [cfg["model"]] * C
  assert len(models) == C
  cameras = [_make_camera(m, rng, focal=1000.0 if ring else 2250.0) for m in models]

  # --- camera poses (rig -> camera) ------------------------------------------------------------
  cam_poses = np.tile(np.eye(4), (C, 1, 1))
  if ring:
    for c in range(C):
      yaw = 2 * np.pi * c / C
      fwd = np.array([np.sin(yaw), 0.0, np.cos(yaw)])
      cam_in_rig = np.eye(4)
      cam_in_rig[:3, :3] = look_rotation(fwd)
      cam_in_rig[:3, 3] = 0.1 * fwd
      cam_poses[c] = np.linalg.inv(cam_in_rig)
    cam_poses = perturb(cam_poses, rng, 0.02, 0.0)
  else:
    for c in range(C):
      d = np.concatenate([rng.normal(0, 0.05, 3), [-0.15 * (c - (C - 1) / 2), 0.0, 0.0]])
      cam_poses[c] = to_matrix(d)

  # --- board poses (board -> world) -------------------------------------------------------------
  board_poses = np.tile(np.eye(4), (B, 1, 1))
  if cfg.get("cube"):
    board_poses = _cube_board_poses(B, 0.4)
    board_poses = np.linalg.inv(board_poses[0]) @ board_poses
  elif ring:
    for b in range(B):
      yaw = 2 * np.pi * b / B
      out = np.array([np.sin(yaw), 0.0, np.cos(yaw)])
      Rb = look_rotation(out)       # board z axis points outward -> board seen from the inside
      m = np.eye(4)
      m[:3, :3] = Rb
      m[:3, 3] = 1.0 * out - Rb @ np.array([0.2, 0.2, 0.0])
      board_poses[b] = m
  elif cfg.get("board_grid"):
    # boards on a planar grid (board_grid columns, 0.42 m pitch) whose middle lies where a single board would be
    cols = cfg["board_grid"]
    rows = (B + cols - 1) // cols
    for b in range(B):
      off = [0.42 * (b % cols - (cols - 1) / 2), 0.42 * (b // cols - (rows - 1) / 2), 0.02 * (b % 3)]
      board_poses[b] = to_matrix(np.concatenate([rng.normal(0, 0.05, 3), off]))
  else:
    for b in range(1, B):
      d = np.concatenate([rng.normal(0, 0.05, 3), [0.45 * b, 0.05 * b, 0.02 * b]])
      board_poses[b] = to_matrix(d)

  # --- rig poses (world -> rig) ----------------------------------------------------------------
  if ring:
    rig = np.tile(np.eye(4), (F, 1, 1))
    for f in range(F):
      yaw = rng.uniform(0, 2 * np.pi)
      d = np.concatenate([[0, yaw, 0], rng.normal(0, 0.05, 3)])
      rig[f] = to_matrix(np.concatenate([rng.normal(0, 0.05, 3), [0, 0, 0]])) @ to_matrix(d)
  else:
    centre = np.array([-0.2, -0.25, 1.0]) if not cfg.get("cube") else np.array([-0.2, -0.2, 1.1])
    if cfg.get("board_grid"):   # look at the middle of the grid from `distance`
      centre = np.array([-0.2, -0.25, cfg.get("distance", 1.0)])
    d = np.concatenate([rng.normal(0, 0.25, (F, 3)), centre + rng.normal(0, 0.1, (F, 3))], axis=1)
    rig = to_matrix(d)

  motion = cfg["motion"]
  rig_end = None
  hand_eye = None
  if motion == "rolling":
    rig_end = to_matrix(np.concatenate([rng.normal(0, 5e-3, (F, 3)), rng.normal(0, 5e-3, (F, 3))], axis=1)) @ rig
  if motion == "hand_eye":
    # rig[f] = gripper_wrt_camera @ base_wrt_gripper[f] @ world_wrt_base   (motion/hand_eye.py:43-46)
    gripper_wrt_camera = to_matrix(np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.1, 3)]))
    world_wrt_base = to_matrix(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.5, 3)]))
    base_wrt_gripper = np.linalg.inv(gripper_wrt_camera) @ rig @ np.linalg.inv(world_wrt_base)
    hand_eye = SimpleNamespace(base_wrt_gripper=base_wrt_gripper, world_wrt_base=world_wrt_base,
                               gripper_wrt_camera=gripper_wrt_camera)

  # --- observations ---------------------------------------------------------------------------
  padded = np.zeros((B, P, 3))
  pvalid = np.zeros((B, P), dtype=bool)
  for b, pts in enumerate(board_pts):
    padded[b, :pts.shape[0]] = pts.astype(np.float64)
    pvalid[b, :pts.shape[0]] = True

  W = np.einsum('bij,bpj->bpi', board_poses[:, :3, :3], padded) + board_poses[:, None, :3, 3]  # [B,P,3]
  normals = board_poses[:, :3, 2]  # board +z axis in world

  points = np.zeros((C, F, B, P, 2))
  valid = np.zeros((C, F, B, P), dtype=bool)
  w, h = IMAGE_SIZE
  fa, fb = (0, F) if obs_frames is None else obs_frames
  # observations are drawn frame by frame from generators seeded with (seed, frame): a frame's data does not depend
  # on which other frames are generated, so every rank of a frame-sharded run can synthesise just its own shard
  for f in range(fa, fb):
    frng = np.random.default_rng([cfg["seed"], 7919, f])
    view_vis = frng.random((C, B)) < 0.7
    point_vis = frng.random((C, B, P)) < 0.9
    gauss = frng.normal(0, noise, (C, B, P, 2))
    for c in range(C):
      T0 = cam_poses[c] @ rig[f]
      X0 = np.einsum('ij,bpj->bpi', T0[:3, :3], W) + T0[:3, 3]
      if motion == "rolling":
        T1 = cam_poses[c] @ rig_end[f]
        X1 = np.einsum('ij,bpj->bpi', T1[:3, :3], W) + T1[:3, 3]
        uv = _project(cameras[c], X0)
        for _ in range(6):  # fixed point for the scan time of the observed row
          t = np.clip(uv[..., 1] / h, 0.0, 1.0)[..., None]
          Xc = X0 * (1 - t) + X1 * t
          uv = _project(cameras[c], Xc)
        # the reference derives t from the observed (noisy) y; make the data consistent with that model
        obs = uv + gauss[c]
        t = (obs[..., 1] / h)[..., None]
        Xc = X0 * (1 - t) + X1 * t
        uv = _project(cameras[c], Xc)
      else:
        Xc = X0
        uv = _project(cameras[c], Xc)
      obs_c = uv + gauss[c]

      nz = np.einsum('ij,bj->bi', T0[:3, :3], normals)            # board normal in camera frame
      facing = np.einsum('bi,bpi->bp', nz, Xc) > 0                  # seen from the printed side
      if not (ring or cfg.get("cube")):
        facing = np.ones_like(facing)
      ok = (Xc[..., 2] > 0.1) & (uv[..., 0] >= 0) & (uv[..., 0] < w) & (uv[..., 1] >= 0) & (uv[..., 1] < h)
      ok &= np.isfinite(uv).all(axis=-1) & facing & pvalid
      if cameras[c].model == 'fisheye':
        ok &= np.arctan2(np.hypot(Xc[..., 0], Xc[..., 1]), Xc[..., 2]) < np.deg2rad(75)
      else:
        ok &= np.hypot(Xc[..., 0], Xc[..., 1]) < 0.62 * Xc[..., 2]  # stay inside the monotone range of the radial model
      if motion == "rolling":
        # both ends of the scan must see the point too: otherwise the scan-time fixed point above can run away
        # (t >> 1) and produce an observation that is inconsistent with t = y_observed / height
        for Xq in (X0, X1):
          ok &= (Xq[..., 2] > 0.1) & (np.hypot(Xq[..., 0], Xq[..., 1]) < 0.62 * Xq[..., 2])
      v = ok & view_vis[c][..., None] & point_vis[c]
      # a detector reports a board only when enough corners are found (charuco.py:99-101)
      v &= (v.sum(axis=-1) >= 12)[..., None]
      points[c, f] = np.where(v[..., None], obs_c, 0.0)
      valid[c, f] = v

    n_out = int(round(outlier_frac * valid[:, f].sum()))
    if n_out > 0:
      vf = valid[:, f]
      idx = np.flatnonzero(vf.ravel())
      pick = frng.choice(idx, size=n_out, replace=False)
      ang = frng.uniform(0, 2 * np.pi, n_out)
      mag = frng.uniform(5, 50, n_out)
      pf = points[:, f].reshape(-1, 2).copy()
      pf[pick] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)
      points[:, f] = pf.reshape(points[:, f].shape)

  truth = SimpleNamespace(cameras=cameras, camera_poses=cam_poses, board_poses=board_poses, rig=rig,
                          rig_end=rig_end, hand_eye=hand_eye)

  # --- initial guess ----------------------------------------------------------------------------
  init_cams = [_perturb_camera(cam, rng) for cam in cameras]
  init_cam_poses = perturb(cam_poses, rng, 0.01, 0.005)
  if motion == "hand_eye":
    # HandEyeCalibration.initialise disables camera_poses and cameras (optimization/hand_eye.py:35):
    # they come from an earlier calibration, so start them at the truth.
    init_cams, init_cam_poses = cameras, cam_poses
  init = SimpleNamespace(
    cameras=init_cams,
    camera_poses=init_cam_poses,
    board_poses=perturb(board_poses, rng, 0.01, 0.005),
    rig=perturb(rig, rng, 0.01, 0.005),
    rig_end=None if rig_end is None else perturb(rig_end, rng, 0.01, 0.005),
    hand_eye=None if hand_eye is None else SimpleNamespace(
      base_wrt_gripper=hand_eye.base_wrt_gripper,
      world_wrt_base=perturb(hand_eye.world_wrt_base, rng, 0.01, 0.005),
      gripper_wrt_camera=perturb(hand_eye.gripper_wrt_camera, rng, 0.01, 0.005)))

  return SimpleNamespace(
    name=name_or_cfg if isinstance(name_or_cfg, str) else "custom", cfg=cfg,
    truth=truth, init=init, board_points=board_pts,
    points=points, valid=valid,
    camera_valid=np.ones(C, dtype=bool), board_valid=np.ones(B, dtype=bool), frame_valid=np.ones(F, dtype=bool),
    optimize=dict(cameras=cfg["optimize_cameras"], boards=False, camera_poses=motion != "hand_eye",
                  board_poses=True, motion=True))


# ------------------------------------------------------------------------------------------------
# (de)serialisation to flat arrays -- used for tests/golden/*.npz
# ------------------------------------------------------------------------------------------------
def _pose_set_arrays(prefix, ps, out):
  out[prefix + "camera_poses"] = ps.camera_poses
  out[prefix + "board_poses"] = ps.board_poses
  out[prefix + "rig"] = ps.rig
  if ps.rig_end is not None:
    out[prefix + "rig_end"] = ps.rig_end
  if ps.hand_eye is not None:
    out[prefix + "he_base_wrt_gripper"] = ps.hand_eye.base_wrt_gripper
    out[prefix + "he_world_wrt_base"] = ps.hand_eye.world_wrt_base
    out[prefix + "he_gripper_wrt_camera"] = ps.hand_eye.gripper_wrt_camera
  out[prefix + "K"] = np.stack([c.intrinsic for c in ps.cameras])
  sizes = [c.dist.size for c in ps.cameras]
  if len(set(sizes)) == 1:
    out[prefix + "dist"] = np.stack([c.dist for c in ps.cameras])
  else:   # cameras of different models: rows padded with zeros, true sizes alongside
    out[prefix + "dist"] = np.stack([np.concatenate([c.dist, np.zeros(max(sizes) - c.dist.size)]) for c in ps.cameras])
    out[prefix + "dist_sizes"] = np.array(sizes)


def rig_to_arrays(rig):
  import json
  out = dict(points=rig.points, valid=rig.valid, camera_valid=rig.camera_valid, board_valid=rig.board_valid,
             frame_valid=rig.frame_valid)
  for i, b in enumerate(rig.board_points):
    out[f"board_points_{i}"] = b
  _pose_set_arrays("init_", rig.init, out)
  _pose_set_arrays("truth_", rig.truth, out)
  cams = rig.init.cameras
  meta = dict(name=rig.name, cfg=rig.cfg, optimize=rig.optimize, n_boards=len(rig.board_points),
              cameras=[dict(model=c.model, image_size=list(c.image_size), fix_aspect=bool(c.fix_aspect),
                            has_skew=bool(c.has_skew)) for c in cams])
  out["meta_json"] = np.array(json.dumps(meta))
  return out


def rig_from_arrays(arrs):
  import json
  meta = json.loads(str(arrs["meta_json"]))

  def pose_set(prefix):
    nds = arrs[prefix + "dist_sizes"] if prefix + "dist_sizes" in arrs else None
    cams = [SimpleNamespace(model=m["model"], image_size=tuple(m["image_size"]), intrinsic=arrs[prefix + "K"][i],
                            dist=arrs[prefix + "dist"][i] if nds is None else arrs[prefix + "dist"][i][:int(nds[i])],
                            fix_aspect=m["fix_aspect"], has_skew=m["has_skew"])
            for i, m in enumerate(meta["cameras"])]
    he = None
    if prefix + "he_world_wrt_base" in arrs:
      he = SimpleNamespace(base_wrt_gripper=arrs[prefix + "he_base_wrt_gripper"],
                           world_wrt_base=arrs[prefix + "he_world_wrt_base"],
                           gripper_wrt_camera=arrs[prefix + "he_gripper_wrt_camera"])
    return SimpleNamespace(cameras=cams, camera_poses=arrs[prefix + "camera_poses"],
                           board_poses=arrs[prefix + "board_poses"], rig=arrs[prefix + "rig"],
                           rig_end=arrs[prefix + "rig_end"] if prefix + "rig_end" in arrs else None, hand_eye=he)

  return SimpleNamespace(
    name=meta["name"], cfg=meta["cfg"], truth=pose_set("truth_"), init=pose_set("init_"),
    board_points=[arrs[f"board_points_{i}"] for i in range(meta["n_boards"])],
    points=arrs["points"], valid=arrs["valid"], camera_valid=arrs["camera_valid"],
    board_valid=arrs["board_valid"], frame_valid=arrs["frame_valid"], optimize=meta["optimize"])


# ------------------------------------------------------------------------------------------------
# per-view board poses for the initialisation tables (tables.make_pose_table would obtain them from cv2.solvePnP)
# ------------------------------------------------------------------------------------------------
def view_pose_errors(rig, poses):
  """Reprojection error of the per-view board poses [C, F, B] against the rig's detections: the RMS distance (px) that
  board.estimate_pose_points reports to tables.extract_pose (tables.py:44-46), with the true cameras standing in for the
  intrinsic calibration that precedes it.  Views without detections get 0."""
  C, F, B, P = rig.valid.shape
  padded = np.zeros((B, P, 3))
  for b, pts in enumerate(rig.board_points):
    padded[b, :pts.shape[0]] = pts
  err = np.zeros((C, F, B))
  for c in range(C):
    X = np.einsum('fbij,bpj->fbpi', poses[c, :, :, :3, :3], padded) + poses[c, :, :, None, :3, 3]
    d2 = ((_project(rig.truth.cameras[c], X) - rig.points[c]) ** 2).sum(axis=-1)
    v = rig.valid[c]
    n = v.sum(axis=-1)
    with np.errstate(all='ignore'):
      err[c] = np.where(n > 0, np.sqrt(np.where(v, d2, 0.0).sum(axis=-1) / np.maximum(n, 1)), 0.0)
  return np.nan_to_num(err, nan=np.inf)


def make_pose_table(rig, seed=0, rot_sigma=None, trans_sigma=None, outlier_frac=0.03, pose_error_limit=None):
  """Pose table [C, F, B] of the rig: truth chain camera . rig . board perturbed by a small SE(3) noise (a PnP estimate
  from noisy corners), a fraction of gross outliers (wrong board orientation), validity / detection counts from the rig's
  observation table.  Returns dict(poses, valid, num_points) of numpy arrays; invalid entries are the identity like
  tables.invalid_pose (tables.py:38).

  pose_error_limit (px; default: the rig configuration's, BASELINE configs[4] "outlier pose rejection" uses the reference's
  1.0): views whose pose reprojects its detections worse than the limit are dropped from the table -- invalid, identity,
  zero points -- as tables.extract_pose does with exclude_bad_poses (tables.py:48-56, workspace.py:196-198).  With the rig
  generator's 1 % gross corner outliers about half of the views of cfg5 go: every view that holds such a corner, and the
  wrong-orientation poses."""
  rng = np.random.default_rng([seed, 4241])
  tr = rig.truth
  C, F, B, P = rig.valid.shape
  chain = tr.camera_poses[:, None, None] @ tr.rig[None, :, None] @ tr.board_poses[None, None, :]
  noise = rig.cfg.get("pose_noise", (2e-3, 1e-3))   # (rad, m)
  poses = perturb(chain, rng, noise[0] if rot_sigma is None else rot_sigma, noise[1] if trans_sigma is None else trans_sigma)
  num_points = rig.valid.sum(axis=3)
  valid = num_points > 0
  out = rng.random((C, F, B)) < outlier_frac
  bad = to_matrix(np.concatenate([rng.normal(0, 0.8, (C, F, B, 3)), rng.normal(0, 0.3, (C, F, B, 3))], axis=-1))
  poses = np.where(out[..., None, None], bad @ poses, poses)
  if pose_error_limit is None:
    pose_error_limit = rig.cfg.get("pose_error_limit")
  rejected = np.zeros_like(valid)
  if pose_error_limit is not None:
    rejected = valid & (view_pose_errors(rig, poses) > pose_error_limit)
    valid = valid & ~rejected
  poses = np.where(valid[..., None, None], poses, np.eye(4))
  return dict(poses=poses, valid=valid, num_points=np.where(valid, num_points, 0), rejected=rejected)
