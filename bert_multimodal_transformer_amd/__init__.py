"""MI355X-native MAG-BERT training hot path (gfx950 HIP kernels behind the reference's Python surfaces).

    from bert_multimodal_transformer_amd import MAG, MAG_BertForSequenceClassification, MultimodalConfig, BertConfig
    from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
    from bert_multimodal_transformer_amd import multimodal_driver

The compute path is libmagbert_hip.so (include/magbert_hip.h, csrc/*.hip); Python is the drop-in boundary
(/root/reference modeling.py / bert.py / multimodal_driver.py signatures) + plumbing (memory, streams, RCCL).
There is no CPU fallback: operators raise if the library or a ROCm device is missing.
"""
from . import _lib
from .global_configs import ACOUSTIC_DIM, VISUAL_DIM, TEXT_DIM
from .modeling import MAG
from .bert import BertConfig, MAG_BertModel, MAG_BertForSequenceClassification
from .xlnet import XLNetConfig, MAG_XLNetModel, MAG_XLNetForSequenceClassification
from .optimization import AdamW, get_linear_schedule_with_warmup
from .multimodal_driver import MultimodalConfig

__all__ = ["MAG", "MAG_BertModel", "MAG_BertForSequenceClassification", "BertConfig", "XLNetConfig", "MAG_XLNetModel", "MAG_XLNetForSequenceClassification", "MultimodalConfig", "AdamW",
           "get_linear_schedule_with_warmup", "ACOUSTIC_DIM", "VISUAL_DIM", "TEXT_DIM"]
__version__ = "0.1.0"
