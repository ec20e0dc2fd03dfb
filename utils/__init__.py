"""Drop-in module names of the reference (`utils.detection_util`, `utils.common`): thin
re-exports of mcm_amd so `from utils.detection_util import get_ood_scores_clip` keeps working."""
