"""espresso_amd — MI355X (gfx950) native hot path of freewym/espresso.

Importing the package registers every component under the reference's names (like
espresso/__init__.py:6-12 does by side effect): tasks, models, criterions, optimizers,
lr schedulers and audio feature transforms.  Compute is hand-written HIP behind a C ABI
(include/espresso_amd.h); see DESIGN.md."""
from . import registry  # noqa: F401
from .criterions import cross_entropy_v2 as _cev2, ctc_loss as _ctc, label_smoothed_cross_entropy_v2 as _lsce  # noqa: F401
from .criterions import cross_entropy as _ce, transducer_loss as _rnnt  # noqa: F401
from .data import feature_transforms as _ft  # noqa: F401
from .models.transformer import speech_transformer_base as _encdec, speech_transformer_encoder_model as _enc_model  # noqa: F401
from .models.transformer import speech_transformer_legacy as _legacy, speech_transformer_transducer_base as _transducer  # noqa: F401
from .models import lstm_lm as _lstm_lm, speech_lstm as _speech_lstm  # noqa: F401
from .optim import adam as _adam, lr_schedulers as _lrs, noam_lr_scheduler as _noam  # noqa: F401
from .tasks import language_modeling_for_asr as _lm_task, speech_recognition as _task  # noqa: F401

__version__ = "0.1.0"
