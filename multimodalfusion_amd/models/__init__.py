from .model_modules import Attn_Net, Attn_Net_Gated, SNN_Block, XlinearFusion, Highway, Residual, ResidualBlock  # noqa: F401
from .model_attention_mil_path import MIL_Attention_fc_path, MIL_Attention_fc_surv_path  # noqa: F401
from .model_attention_mil_radio import MIL_Attention_fc_radio, MIL_Attention_fc_surv_radio  # noqa: F401
from .model_genomic import MaxNet, MaxNet_base  # noqa: F401
from .model_mm_attention_mil import MM_MIL_Attention_fc, MM_MIL_Attention_fc_surv  # noqa: F401
from . import nll_models_pretrained, coxranking_models_pretrained  # noqa: F401,E402  (stage 2: class names collide, import the modules)
