"""Checkpoint save / resume, in the spirit of /root/reference/main.py:356-427 (restore) and
:469-519 (save): files are named `checkpoint-<best_score digits>-<global_t>` (`str(abs(best_score))[2:8]`,
main.py:495-502), `global_t` is parsed back from the file name (:404-411), wall time lives in
`wall_t.<global_t>` (:416-419, 481-487), at most 20 checkpoints are kept (:356), and the next save step is
rounded up to a multiple of save_interval_step (:418).

The payload is the flat fp32 parameter buffer (TF layouts, TF variable order -- see model.param_spec) plus the
per-variable offset table, so it can be converted to / from a TF checkpoint of the reference by name.  The
reference's Saver does NOT include the RMSProp slots (they are created after the saved-variable list is taken,
SURVEY 5.4) and restarts them at rms = 1, momentum = 0; here they are saved too and `restore(..., restore_slots=
False)` reproduces the reference's behaviour.  Replay, LSTM state and RNG are not saved (reference: neither)."""
import glob
import os
import warnings

import numpy as np
import torch

MAX_TO_KEEP = 20


def checkpoint_name(best_score, global_t, name=""):
    """`<name>-<str(abs(best_score))[2:8]>-<global_t>.pt` (main.py:495-502).  The score field is whatever that slice
    yields -- '123456' for -0.123456789, '.5' for 12.5, '0.0' for 100.0, '-05' for 1e-05 -- and is never parsed back."""
    base = name if name else "checkpoint"
    return "%s-%s-%d.pt" % (base, str(abs(float(best_score)))[2:8], int(global_t))


def _parse_global_t(fn, base):
    """global_t of `<base>-<anything>-<global_t>.pt`, parsed like the reference does (main.py:401-411: split on '-', take
    the step token); None when the tail after the last '-' is not a step count (another run family, e.g.
    'checkpoint-best-final.pt')."""
    rest = fn[len(base) + 1:-len(".pt")]
    if "-" not in rest:
        return None
    tail = rest.rsplit("-", 1)[1]
    return int(tail) if tail.isdigit() else None


def list_checkpoints(checkpoint_dir, name=""):
    """[(global_t, path)] of the checkpoints called `name` (default "checkpoint"), sorted by global_t.  Anything else in the
    directory (best.pt, an exported model, another prefix, a file of the prefix without a step count) is not ours: it is
    left alone with a warning, like the reference, which only looks at what its Saver wrote (main.py:382-411)."""
    base = name if name else "checkpoint"
    out = []
    for p in sorted(glob.glob(os.path.join(checkpoint_dir, "*.pt"))):
        fn = os.path.basename(p)
        t = _parse_global_t(fn, base) if fn.startswith(base + "-") else None
        if t is None:
            warnings.warn("%s: not a '%s-<score>-<global_t>.pt' checkpoint, ignored" % (p, base))
            continue
        out.append((t, p))
    return sorted(out)


def next_save_steps(global_t, save_interval_step):
    return (global_t + save_interval_step) // save_interval_step * save_interval_step


def save(checkpoint_dir, net, applier, global_t, wall_t, best_score=0.0, name=""):
    os.makedirs(checkpoint_dir, exist_ok=True)
    with open(os.path.join(checkpoint_dir, "wall_t." + str(int(global_t))), "w") as f:
        f.write(str(float(wall_t)))
    payload = {
        "format": "unreal_amd.flat.v1",
        "global_t": int(global_t),
        "best_score": float(best_score),
        "spec": [(n, tuple(s)) for n, s, _ in net.spec],
        "offsets": {k: (o, n) for k, (o, n, _) in net.params.offsets.items()},
        "params": net.params.flat.detach().cpu(),
        "rms": None if applier is None or applier.ms is None else applier.ms.detach().cpu(),
        "momentum": None if applier is None or applier.mom is None else applier.mom.detach().cpu(),
    }
    path = os.path.join(checkpoint_dir, checkpoint_name(best_score, global_t, name))
    torch.save(payload, path)
    ck = list_checkpoints(checkpoint_dir, name)
    for t, p in ck[:-MAX_TO_KEEP]:
        os.remove(p)
        w = os.path.join(checkpoint_dir, "wall_t." + str(t))
        if os.path.exists(w):
            os.remove(w)
    return path


def restore(checkpoint_dir, net, applier=None, restore_slots=True, name=""):
    """-> (global_t, wall_t, best_score) of the newest checkpoint, or None if there is none."""
    ck = list_checkpoints(checkpoint_dir, name)
    if not ck:
        return None
    global_t, path = ck[-1]
    payload = torch.load(path, map_location="cpu", weights_only=True)
    if [(n, tuple(s)) for n, s in payload["spec"]] != [(n, tuple(s)) for n, s, _ in net.spec]:
        raise ValueError("checkpoint %s holds a different variable list than this model" % path)
    net.params.flat.copy_(payload["params"])
    if hasattr(net, "mark_params_changed"):      # derived weight planes must follow the restored parameters
        net.mark_params_changed()
    if applier is not None:
        applier._create_slots(net.params.flat)
        if restore_slots and payload.get("rms") is not None:
            applier.ms.copy_(payload["rms"])
            applier.mom.copy_(payload["momentum"])
        else:                                   # the reference's behaviour: slots restart at 1 / 0
            applier.ms.fill_(1.0)
            applier.mom.zero_()
    wall = os.path.join(checkpoint_dir, "wall_t." + str(global_t))
    wall_t = float(open(wall).read()) if os.path.exists(wall) else 0.0
    assert payload["global_t"] == global_t
    return global_t, wall_t, payload.get("best_score", 0.0)


def export_npz(path, net):
    """Named TF-layout arrays ({variable name: array}) for interchange with the reference's variables."""
    np.savez_compressed(path, **net.export_named())


def import_npz(path, net):
    z = np.load(path, allow_pickle=False)
    net.load_named({k: z[k] for k in z.files})
