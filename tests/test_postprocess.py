"""ocrs_models_amd/postprocess.py: the detection validation metrics of the reference's ``test()`` (postprocess.py:11-36, 102-187) restated
without OpenCV / shapely.  Neither library exists here, so these are known-answer and property tests (parity with cv2 is unpinned and says
so in the module): rectangles with known corners / areas / IoUs, rotation invariance, the matching rules on constructed scenes."""
import math

import numpy as np
import torch

from ocrs_models_amd import postprocess as pp


def _rect_mask(H, W, boxes):
    m = torch.zeros(H, W, dtype=torch.uint8)
    for x0, y0, x1, y1 in boxes:
        m[y0:y1 + 1, x0:x1 + 1] = 1
    return m


def _sorted_pts(q):
    return sorted((round(float(x), 4), round(float(y), 4)) for x, y in q)


def test_axis_aligned_components_give_their_pixel_centre_boxes():
    # cv2.minAreaRect works on pixel-centre coordinates: a block covering x 3..10, y 5..8 has corners (3,5) (10,5) (10,8) (3,8)
    boxes = [(3, 5, 10, 8), (20, 2, 22, 30), (40, 40, 40, 40)]
    q = pp.extract_cc_quads(_rect_mask(64, 64, boxes))
    assert q.shape == (3, 4, 2)
    got = sorted(_sorted_pts(x) for x in q)
    want = sorted(_sorted_pts([(x0, y0), (x1, y0), (x1, y1), (x0, y1)]) for x0, y0, x1, y1 in boxes)
    assert got == want
    assert pp.extract_cc_quads(torch.zeros(1, 8, 8)).shape == (0, 4, 2)


def test_diagonal_touching_pixels_are_one_component_and_quads_are_minimal():
    m = torch.zeros(16, 16, dtype=torch.uint8)
    for i in range(2, 12):
        m[i, i] = 1                    # a diagonal line: 8-connected -> ONE component, zero-area rectangle along the diagonal
    q = pp.extract_cc_quads(m)
    assert q.shape[0] == 1
    assert pp._area(q[0].numpy().astype(np.float64)) < 1e-6
    # a rotated rectangle drawn as a filled polygon: the quad's area is within a pixel ring of the true one and never smaller than the hull
    yy, xx = np.mgrid[0:120, 0:120]
    c, s = math.cos(0.5), math.sin(0.5)
    u, v = (xx - 60) * c + (yy - 60) * s, -(xx - 60) * s + (yy - 60) * c
    m2 = torch.from_numpy(((np.abs(u) <= 40) & (np.abs(v) <= 10)).astype(np.uint8))
    q2 = pp.extract_cc_quads(m2)[0].numpy().astype(np.float64)
    a = pp._area(q2)
    assert 0.93 * 80 * 20 < a < 1.05 * 80 * 20, a
    ys, xs = np.nonzero(m2.numpy())
    # every foreground pixel centre lies inside the quad (it is an enclosing rectangle)
    inside = pp._clip(np.array([[0.0, 0.0]]), pp._ccw(q2))  # (smoke of the clipper on a point outside)
    assert len(inside) == 0
    e = np.roll(pp._ccw(q2), -1, axis=0) - pp._ccw(q2)
    for (ax, ay), (ex, ey) in zip(pp._ccw(q2), e):
        assert np.all(ex * (ys - ay) - ey * (xs - ax) >= -1e-3)


def test_intersection_areas_known_answers_and_symmetry():
    sq = lambda x, y, s: np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s]], dtype=np.float64)  # noqa: E731
    assert abs(pp.quad_intersection_area(sq(0, 0, 4), sq(2, 2, 4)) - 4.0) < 1e-12
    assert abs(pp.quad_intersection_area(sq(0, 0, 4), sq(1, 1, 2)) - 4.0) < 1e-12       # containment
    assert pp.quad_intersection_area(sq(0, 0, 4), sq(4, 0, 4)) == 0.0                    # shared edge only
    assert pp.quad_intersection_area(sq(0, 0, 4), sq(10, 10, 1)) == 0.0
    # a unit square and the same square rotated by 45 degrees about its centre intersect in a regular octagon of area 2(sqrt 2 - 1)
    c = math.sqrt(0.5)
    diamond = np.array([[0.5, 0.5 - c], [0.5 + c, 0.5], [0.5, 0.5 + c], [0.5 - c, 0.5]])
    assert abs(pp.quad_intersection_area(sq(0, 0, 1), diamond) - 2 * (math.sqrt(2) - 1)) < 1e-12
    # symmetry, orientation independence, rotation invariance
    r = np.random.RandomState(0)
    for _ in range(50):
        def rnd():
            cx, cy, w, h, t = r.uniform(-3, 3), r.uniform(-3, 3), r.uniform(0.5, 4), r.uniform(0.5, 4), r.uniform(0, math.pi)
            R = np.array([[math.cos(t), -math.sin(t)], [math.sin(t), math.cos(t)]])
            return (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + [cx, cy]
        a, b = rnd(), rnd()
        i1, i2, i3 = pp.quad_intersection_area(a, b), pp.quad_intersection_area(b, a), pp.quad_intersection_area(a[::-1], b)
        assert abs(i1 - i2) < 1e-9 and abs(i1 - i3) < 1e-9
        t = 0.7
        R = np.array([[math.cos(t), -math.sin(t)], [math.sin(t), math.cos(t)]])
        assert abs(pp.quad_intersection_area(a @ R.T, b @ R.T) - i1) < 1e-9
        assert -1e-12 <= i1 <= min(pp._area(a), pp._area(b)) + 1e-9


def test_box_match_metrics_rules():
    box = lambda x0, y0, x1, y1: [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]  # noqa: E731
    T = torch.tensor([box(0, 0, 10, 4), box(20, 0, 30, 4), box(0, 10, 10, 14), box(20, 10, 30, 14)], dtype=torch.float32)
    # identical sets: everything matches
    assert pp.box_match_metrics(T, T) == {"precision": 1.0, "recall": 1.0, "merged_frac": 0.0, "split_frac": 0.0}
    # empty cases follow the reference's conventions
    assert pp.box_match_metrics(torch.zeros(0, 4, 2), T) == {"precision": 1.0, "recall": 0.0, "merged_frac": 0.0, "split_frac": 0.0}
    assert pp.box_match_metrics(T, torch.zeros(0, 4, 2)) == {"precision": 0.0, "recall": 1.0, "merged_frac": 0.0, "split_frac": 0.0}
    # one prediction covering the two upper targets (merged), the third target found (IoU 0.8), the fourth split into two halves
    P = torch.tensor([box(0, 0, 30, 4), box(0, 10, 8, 14), box(20, 10, 25, 14), box(25, 10, 30, 14)], dtype=torch.float32)
    m = pp.box_match_metrics(P, T)
    assert m["precision"] == 1 / 4 and m["recall"] == 1 / 4
    assert m["merged_frac"] == 2 / 4 and m["split_frac"] == 1 / 4
    # IoU exactly 0.5 is not a match (strict threshold)
    assert pp.box_match_metrics(torch.tensor([box(0, 0, 10, 2)], dtype=torch.float32), torch.tensor([box(0, 0, 10, 4)], dtype=torch.float32))["recall"] == 0.0


def test_mask_metrics_end_to_end_and_validation_loop_returns_them():
    tgt = _rect_mask(64, 96, [(4, 4, 30, 10), (40, 4, 80, 10), (4, 30, 50, 38)])
    pred = _rect_mask(64, 96, [(4, 4, 30, 10), (40, 5, 80, 10), (60, 50, 70, 55)])   # two good matches, one miss, one false positive
    m = pp.mask_metrics(pred, tgt)
    assert m["precision"] == 2 / 3 and m["recall"] == 2 / 3 and m["merged_frac"] == 0.0 and m["split_frac"] == 0.0
    from ocrs_models_amd.train_detection import get_metric_means
    assert set(get_metric_means([m, m])) == {"precision", "recall", "merged_frac", "split_frac"}
