"""Word-level validation metrics of the detection ``test()`` loop (reference: ocrs_models/postprocess.py:11-36 ``extract_cc_quads``,
:102-187 ``box_match_metrics``; used by train_detection.py:177-184), without OpenCV / shapely (neither is installed where this is built).

Host-side numpy, like the reference's own CPU post-processing -- this is not part of the accelerated path.  What is restated:

* ``extract_cc_quads``: the reference takes ``cv2.findContours(RETR_EXTERNAL)`` and ``cv2.boxPoints(cv2.minAreaRect(contour))``.  The
  minimum-area rectangle of an outer contour is the minimum-area rectangle of the convex hull of the component's pixel centres, so the
  restatement labels the 8-connected components (``scipy.ndimage.label``), takes each component's hull (monotone chain) and runs rotating
  calipers over the hull edges.  Differences that cannot be pinned without OpenCV (parity UNPINNED, like the resize of the input pipeline):
  the vertex ORDER of a quad (irrelevant to the metrics), the choice among equal-area rectangles, and components that lie inside a hole of
  another component (RETR_EXTERNAL drops them, they are kept here).
* ``box_match_metrics``: the quads are convex, so intersection areas come from Sutherland-Hodgman clipping and union = a + b - intersection
  (shapely's general polygon overlay is not needed); the matching rules and the four reported numbers are the reference's.
"""
from __future__ import annotations

import numpy as np
import torch


def _hull(points: np.ndarray) -> np.ndarray:
    """convex hull of integer points [n][2] (x, y), counter-clockwise, no repeated end point (Andrew's monotone chain)"""
    pts = np.unique(points, axis=0)
    if len(pts) <= 2:
        return pts.astype(np.float64)
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]

    def half(seq):
        out = []
        for p in seq:
            while len(out) >= 2:
                (ax, ay), (bx, by) = out[-2], out[-1]
                if (bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax) <= 0:
                    out.pop()
                else:
                    break
            out.append((int(p[0]), int(p[1])))
        return out

    lower, upper = half(pts), half(pts[::-1])
    return np.array(lower[:-1] + upper[:-1], dtype=np.float64)


def _min_area_rect(hull: np.ndarray) -> np.ndarray:
    """corners [4][2] of the minimum-area enclosing rectangle of a convex polygon (one side is collinear with a hull edge)"""
    n = len(hull)
    if n == 0:
        return np.zeros((4, 2))
    if n == 1:
        return np.repeat(hull, 4, axis=0)
    if n == 2:
        return np.array([hull[0], hull[1], hull[1], hull[0]])
    edges = np.roll(hull, -1, axis=0) - hull
    lens = np.hypot(edges[:, 0], edges[:, 1])
    ux = edges / lens[:, None]                       # unit vectors along each edge
    uy = np.stack([-ux[:, 1], ux[:, 0]], axis=1)     # and their normals
    px = hull @ ux.T                                 # [point][edge] projections
    py = hull @ uy.T
    x0, x1, y0, y1 = px.min(0), px.max(0), py.min(0), py.max(0)
    k = int(np.argmin((x1 - x0) * (y1 - y0)))
    c = lambda a, b: a * ux[k] + b * uy[k]           # noqa: E731
    return np.array([c(x0[k], y0[k]), c(x1[k], y0[k]), c(x1[k], y1[k]), c(x0[k], y1[k])])


def extract_cc_quads(mask: torch.Tensor) -> torch.Tensor:
    """Bounding quads [N][4][2] (x, y) of the connected components of a binary mask (H x W or 1 x H x W): postprocess.py:11-36."""
    from scipy import ndimage

    if mask.dim() > 2:
        if mask.shape[0] != 1:
            raise ValueError("Expected mask to be an HxW or 1xHxW tensor")
        mask = mask[0]
    m = mask.detach().cpu().to(torch.uint8).numpy() != 0
    labels, n = ndimage.label(m, structure=np.ones((3, 3), dtype=bool))  # 8-connectivity, as cv2.findContours traces foreground
    quads = []
    if n:
        ys, xs = np.nonzero(labels)
        lab = labels[ys, xs]
        order = np.argsort(lab, kind="stable")
        ys, xs, lab = ys[order], xs[order], lab[order]
        starts = np.searchsorted(lab, np.arange(1, n + 2))
        for i in range(n):
            sl = slice(starts[i], starts[i + 1])
            quads.append(_min_area_rect(_hull(np.stack([xs[sl], ys[sl]], axis=1))))
    return torch.tensor(np.array(quads, dtype=np.float32).reshape(-1, 4, 2))


def _area(poly: np.ndarray) -> float:
    if len(poly) < 3:
        return 0.0
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _ccw(poly: np.ndarray) -> np.ndarray:
    x, y = poly[:, 0], poly[:, 1]
    return poly if float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) >= 0 else poly[::-1]


def _clip(subject: np.ndarray, clipper: np.ndarray) -> np.ndarray:
    """Sutherland-Hodgman: the part of convex ``subject`` inside convex counter-clockwise ``clipper``"""
    out = [tuple(p) for p in subject]
    for i in range(len(clipper)):
        a, b = clipper[i], clipper[(i + 1) % len(clipper)]
        if not out:
            break
        inp, out = out, []

        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])

        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return np.array(out, dtype=np.float64).reshape(-1, 2)


def quad_intersection_area(a: np.ndarray, b: np.ndarray) -> float:
    """area of the intersection of two convex quads [4][2]"""
    if _area(a) == 0.0 or _area(b) == 0.0:
        return 0.0
    return _area(_clip(_ccw(np.asarray(a, dtype=np.float64)), _ccw(np.asarray(b, dtype=np.float64))))


def box_match_metrics(pred: torch.Tensor, target: torch.Tensor) -> dict[str, float]:
    """precision / recall / merged_frac / split_frac between two sets of rotated rects [N][4][2]: postprocess.py:102-187.

    A prediction is a good match of a target when their IoU exceeds 0.5; targets of which more than half is covered by one prediction
    that covers several are "merged", targets more than half of several predictions lie in are "split"."""
    P = np.asarray(pred.detach().cpu().numpy() if isinstance(pred, torch.Tensor) else pred, dtype=np.float64).reshape(-1, 4, 2)
    T = np.asarray(target.detach().cpu().numpy() if isinstance(target, torch.Tensor) else target, dtype=np.float64).reshape(-1, 4, 2)
    pa = np.array([_area(p) for p in P])
    ta = np.array([_area(t) for t in T])
    inter = np.zeros((len(P), len(T)))
    if len(P) and len(T):
        pmin, pmax, tmin, tmax = P.min(1), P.max(1), T.min(1), T.max(1)
        # the reference's cheap bounding-box test (strict overlap in both axes) decides which pairs get an intersection at all
        cand = ((pmin[:, None, :] < tmax[None, :, :]) & (tmin[None, :, :] < pmax[:, None, :])).all(-1)
        for i, j in zip(*np.nonzero(cand)):
            inter[i, j] = quad_intersection_area(P[i], T[j])
    union = np.where(inter > 0, pa[:, None] + ta[None, :] - inter, 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union                      # 0 / 0 -> nan, which is not > 0.5 (as in the reference)
        cover_t = inter / ta[None, :]            # share of each target inside each prediction
        cover_p = inter / pa[:, None]            # share of each prediction inside each target
    matches = {}
    for i, j in zip(*np.nonzero(iou > 0.5)):     # row-major like torch.nonzero: a later target overwrites an earlier one
        matches[int(i)] = int(j)
    merged = 0
    for i in range(len(P)):
        covered = int(np.count_nonzero(cover_t[i] > 0.5))
        if covered > 1:
            merged += covered
    split = sum(1 for j in range(len(T)) if np.count_nonzero(cover_p[:, j] > 0.5) > 1)
    return {
        "precision": len(matches) / len(P) if len(P) > 0 else 1.0,
        "recall": len(matches) / len(T) if len(T) > 0 else 1.0,
        "merged_frac": merged / len(T) if len(T) > 0 else 0.0,
        "split_frac": split / len(T) if len(T) > 0 else 0.0,
    }


def mask_metrics(bin_pred_mask: torch.Tensor, bin_target_mask: torch.Tensor) -> dict[str, float]:
    """what train_detection.py:177-184 computes per image: quads of both masks, then the box-match metrics"""
    return box_match_metrics(extract_cc_quads(bin_pred_mask), extract_cc_quads(bin_target_mask))
