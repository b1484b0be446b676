"""Consumers of the XYZ image on the reference's path.

    valid_points(points_3D, colors, disparity_map)   main.ipynb:726-737  mask + boolean indexing
    write_point_cloud(path, points, colors)           main.ipynb:795-797  o3d.io.write_point_cloud(.ply)

The compaction runs on the GPU (ordered, so the result equals numpy's `points_3D[mask]`); the
PLY writer is plain host I/O and replaces the Open3D dependency (absent here) for export only --
interactive viewing (o3d.visualization.draw_geometries) is out of scope.
"""
from __future__ import annotations

import numpy as np

from . import stereo as _cv


def valid_points(points_3D, colors, disparity_map):
    """Returns (valid_points float32 (N,3), valid_colors uint8 (N,3) or None) exactly like
    `points_3D[mask]`, `colors[mask]` with mask = ~isnan(X) & ~isinf(X) & (disparity_map > 0)."""
    pts = np.asarray(points_3D)
    disp = np.asarray(disparity_map)
    if pts.ndim != 3 or pts.shape[2] != 3 or pts.shape[:2] != disp.shape:
        raise _cv.error("valid_points: points_3D must be (H, W, 3) and disparity_map (H, W)")
    if colors is not None:
        colors = np.asarray(colors)
        if colors.shape != pts.shape:
            raise _cv.error("valid_points: colors must have the shape of points_3D")
    return _cv.get_engine(_cv._DEFAULT).compact_points_host(pts, disp, colors)


def write_point_cloud(filename, points, colors=None, binary=True) -> bool:
    """PLY export of a point list; `points` (N,3) or (H,W,3), non-finite points are written as they
    are (the notebook saves the unmasked H*W points, main.ipynb:795-797).  Same property layout
    as Open3D's writer: double x/y/z, uchar red/green/blue when colours are given."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    if colors is not None:
        col = np.asarray(colors).reshape(-1, 3)
        if col.shape[0] != n:
            raise _cv.error("write_point_cloud: colors and points differ in length")
        if col.dtype != np.uint8:  # Open3D keeps colours in [0,1]; accept both
            col = np.clip(np.rint(np.asarray(col, np.float64) * (255.0 if col.max(initial=0) <= 1.0 else 1.0)), 0, 255).astype(np.uint8)
    else:
        col = None
    head = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0",
            "comment stereo_reconstruction_cv_amd", f"element vertex {n}",
            "property double x", "property double y", "property double z"]
    if col is not None:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
    head.append("end_header")
    with open(filename, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        if binary:
            if col is None:
                pts.astype("<f8").tofile(f)
            else:
                rec = np.empty(n, dtype=[("p", "<f8", 3), ("c", "u1", 3)])
                rec["p"], rec["c"] = pts, col
                rec.tofile(f)
        else:
            for i in range(n):
                line = " ".join(repr(float(v)) for v in pts[i])
                if col is not None:
                    line += " " + " ".join(str(int(v)) for v in col[i])
                f.write((line + "\n").encode("ascii"))
    return True


def read_point_cloud(filename):
    """Minimal reader for files written by write_point_cloud (used by the tests)."""
    with open(filename, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        n = int([h for h in header if h.startswith("element vertex")][0].split()[-1])
        has_col = any("red" in h for h in header)
        if "format binary_little_endian 1.0" in header:
            dt = [("p", "<f8", 3)] + ([("c", "u1", 3)] if has_col else [])
            rec = np.fromfile(f, dtype=dt, count=n)
            return rec["p"].copy(), (rec["c"].copy() if has_col else None)
        rows = [f.readline().decode("ascii").split() for _ in range(n)]
        pts = np.array([[float(v) for v in r[:3]] for r in rows], np.float64).reshape(n, 3)
        col = np.array([[int(v) for v in r[3:6]] for r in rows], np.uint8).reshape(n, 3) if has_col else None
        return pts, col
