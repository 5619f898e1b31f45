"""Thin wrappers over third-party CPU PnP-RANSAC solvers (used only when pycolmap / OpenCV are installed).
Call signatures follow nerfmatch/utils/geometry.py:189-265; the solvers themselves are out of scope."""
import numpy as np


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def estimate_pose(pts2d, pts3d, K, ransac_thres=1):
    import cv2

    if len(pts2d) < 4:
        return None
    p2, p3, Kn = _np(pts2d).astype(np.float32), _np(pts3d), _np(K)
    ok, rvec, tvec, inl = cv2.solvePnPRansac(p3, p2, cameraMatrix=Kn, distCoeffs=None, reprojectionError=ransac_thres, flags=cv2.SOLVEPNP_AP3P)
    if not ok or np.any(np.isnan(tvec)):
        return None
    inl = inl.ravel()
    rvec, tvec = cv2.solvePnPRefineLM(p3[inl], p2[inl], cameraMatrix=Kn, distCoeffs=None, rvec=rvec, tvec=tvec)
    return cv2.Rodrigues(rvec)[0], tvec.ravel(), inl


def estimate_pose_pycolmap(pts2d, pts3d, K, img_wh=None, ransac_thres=1, center_subpixel=False):
    import pycolmap

    p2, p3, Kn = _np(pts2d), _np(pts3d), _np(K)
    if center_subpixel:
        p2 = p2 + np.array([[0.5, 0.5]], dtype=np.float32)
    if len(p2) < 4:
        return None
    wh = img_wh or (Kn[0, 2] * 2, Kn[1, 2] * 2)
    cam = pycolmap.Camera(model="PINHOLE", width=int(wh[0]), height=int(wh[1]), params=[Kn[0, 0], Kn[1, 1], Kn[0, 2], Kn[1, 2]])
    res = pycolmap.absolute_pose_estimation(p2, p3, cam, max_error_px=ransac_thres)
    if not res["success"]:
        return None
    q = res["qvec"]
    R = np.array([[1 - 2 * q[2] ** 2 - 2 * q[3] ** 2, 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[3] * q[1] + 2 * q[0] * q[2]],
                  [2 * q[1] * q[2] + 2 * q[0] * q[3], 1 - 2 * q[1] ** 2 - 2 * q[3] ** 2, 2 * q[2] * q[3] - 2 * q[0] * q[1]],
                  [2 * q[3] * q[1] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], 1 - 2 * q[1] ** 2 - 2 * q[2] ** 2]])
    return R, res["tvec"], np.where(res["inliers"])[0]
