"""Import the read-only reference (``/root/reference``) in the BUILD container only.

The reference's training scripts import torchvision / wandb / cv2 / shapely /
pylev, none of which are installed here and none of which the hot path touches
(SURVEY.md appendix B).  We register inert stub modules for them, then import
the reference modules.  Used only by ``tools/gen_goldens.py``; nothing under
``tests/`` (gpu or not), ``bench.py`` or the package imports this file.
"""
from __future__ import annotations

import sys
import types


class _Any:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


def _lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def _ga(k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()

    m.__getattr__ = _ga  # type: ignore[attr-defined]
    sys.modules[name] = m
    return m


def import_reference(path="/root/reference"):
    sys.dont_write_bytecode = True
    import torch  # noqa: F401  (import the real torch before any stub is registered)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional")
    _stub("torchvision.io")
    _stub("wandb")
    _stub("cv2")
    _stub("pylev", levenshtein=_lev)
    _stub("shapely")
    _stub("shapely.geometry", MultiLineString=_Any, JOIN_STYLE=_Any())
    _stub("shapely.geometry.polygon", LinearRing=_Any, Polygon=_Any)
    try:
        import tqdm  # noqa: F401
    except ImportError:
        _stub("tqdm", tqdm=lambda x, *a, **k: x)
    try:
        import PIL  # noqa: F401
    except ImportError:
        _stub("PIL", Image=_Any(), ImageDraw=_Any())
        _stub("PIL.Image")
        _stub("PIL.ImageDraw")
    if path not in sys.path:
        sys.path.insert(0, path)
    import ocrs_models.models as models
    import ocrs_models.train_detection as td
    import ocrs_models.train_rec as tr
    from ocrs_models.datasets import util as dutil
    from ocrs_models.datasets.hiertext import DEFAULT_ALPHABET

    return types.SimpleNamespace(models=models, td=td, tr=tr, util=dutil, alphabet=DEFAULT_ALPHABET)
