"""Same import path and signatures as the reference's utils/detection_util.py (:37-45,
:66-119, :209-265); the implementations live in mcm_amd (MI355X-native hot path)."""
from mcm_amd.detection import (encode_prompt_bank, get_and_print_results,  # noqa: F401
                               get_Mahalanobis_score, get_mean_prec, get_ood_scores_clip,
                               print_measures)
from mcm_amd.metrics import fpr_at_recall, get_measures  # noqa: F401
