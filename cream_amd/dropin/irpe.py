"""Drop-in for iRPE/DeiT-with-iRPE/irpe.py (`from irpe import build_rpe`,
rpe_vision_transformer.py:43) — re-export of cream_amd.irpe."""
from cream_amd.irpe import (METHOD, RPEConfig, build_rpe, get_bucket_ids_2d, get_num_buckets,  # noqa: F401
                            get_rpe_config, get_single_rpe_config, iRPE, iRPE_Cross, piecewise_index)
