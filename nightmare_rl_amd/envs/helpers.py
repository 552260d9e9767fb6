"""class_to_dict / get_load_path with the behaviour of reference envs/helpers.py:3-42."""
import os


def class_to_dict(obj) -> dict:
    """Public attributes -> dict, recursively; keys come out in dir() (alphabetical) order, which is what fixes
    the reward evaluation order downstream (reference envs/nightmare_v3_env.py:132-137)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def get_load_path(root, load_run=-1, checkpoint=-1):
    """Latest run directory / latest model_<it>.pt (lexicographic run sort, zero-padded model sort)."""
    try:
        runs = sorted(os.listdir(root))
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = sorted((f for f in os.listdir(load_run) if "model" in f), key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)
