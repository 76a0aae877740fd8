"""Clean-shot detection of the eval=True path (reference models/mpti.py:87-223, 316-371)."""
from . import ops


def shot_keep_flags(model, sfeat_pm, sfeatT, support_x, support_y):
    """(n_way*k_shot) int32 device flags: 0 = the shot's foreground points are ignored when the
    class prototypes are built (the reference's pl_support_y, which is constant within a shot)."""
    return ops.clean_shot_detect(sfeat_pm, support_x, support_y, model.n_way, model.k_shot, model.n_points)
