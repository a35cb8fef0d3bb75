"""ocrs_models_amd -- MI355X-native (gfx950) detection / recognition train-step hot path of
robertknight/ocrs-models: same nn.Module and loss signatures as ocrs_models/models.py,
train_detection.py and train_rec.py, executed by hand-written HIP kernels (libocrs_hip.so)."""
from . import checkpoint, export, graph, input_pipeline, optim, sampler, text  # noqa: F401
from .losses import CTCLoss, balanced_cross_entropy_loss  # noqa: F401
from .models import DetectionModel  # noqa: F401
from .recognition import RecognitionModel  # noqa: F401
from . import losses, train_detection, train_rec  # noqa: F401,E402
