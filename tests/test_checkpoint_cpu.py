"""Checkpoint file protocol (CPU): naming / global_t parsing / wall_t / rotation as in the reference
(main.py:404-419, 481-502), payload round trip, slot-restart option."""
import os

import numpy as np
import torch

from unreal_amd import checkpoint as ck
from unreal_amd.model.model import FlatParams, param_spec


class _Net(object):
    def __init__(self):
        self.spec = param_spec(4)
        self.params = FlatParams(self.spec, "cpu")
        self.params.flat.copy_(torch.arange(self.params.size, dtype=torch.float32) * 1e-6)

    def export_named(self):
        return {n: self.params.shaped(n).numpy().copy() for n, _, _ in self.spec}

    def load_named(self, named):
        for k, v in named.items():
            self.params.views[k].copy_(torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(-1)))


class _Applier(object):
    ms = mom = None

    def _create_slots(self, flat):
        if self.ms is None:
            self.ms, self.mom = torch.ones_like(flat), torch.zeros_like(flat)


def test_names_and_schedule():
    assert ck.checkpoint_name(-0.123456789, 200000) == "checkpoint-123456-200000.pt"
    assert ck.checkpoint_name(0.0, 100) == "checkpoint-0-100.pt"            # str(abs(0.0))[2:8] == '0'
    assert ck.next_save_steps(0, 100000) == 100000 and ck.next_save_steps(250001, 100000) == 300000


def test_round_trip_and_rotation(tmp_path):
    d = str(tmp_path / "ckpt")
    net, app = _Net(), _Applier()
    app._create_slots(net.params.flat)
    app.ms.mul_(0.5)
    app.mom.add_(0.25)
    for k in range(23):
        net.params.flat.add_(1.0)
        ck.save(d, net, app, global_t=(k + 1) * 100000, wall_t=12.5 * (k + 1), best_score=-0.5)
    lst = ck.list_checkpoints(d)
    assert len(lst) == 20 and lst[0][0] == 400000 and lst[-1][0] == 2300000      # max_to_keep = 20
    assert not os.path.exists(os.path.join(d, "wall_t.100000")) and os.path.exists(os.path.join(d, "wall_t.2300000"))
    want = net.params.flat.clone()
    net2, app2 = _Net(), _Applier()
    g, w, s = ck.restore(d, net2, app2)
    assert (g, w, s) == (2300000, 12.5 * 23, -0.5)
    assert torch.equal(net2.params.flat, want) and torch.equal(app2.ms, app.ms) and torch.equal(app2.mom, app.mom)
    net3, app3 = _Net(), _Applier()
    ck.restore(d, net3, app3, restore_slots=False)          # reference behaviour: slots restart
    assert torch.equal(net3.params.flat, want) and float(app3.ms.min()) == 1.0 and float(app3.mom.abs().max()) == 0.0
    assert ck.restore(str(tmp_path / "empty"), net3) is None
    p = str(tmp_path / "vars.npz")
    ck.export_npz(p, net)
    net4 = _Net()
    net4.params.flat.zero_()
    ck.import_npz(p, net4)
    for n, _, _ in net.spec:
        assert torch.equal(net4.params.views[n], net.params.views[n])


def test_unrelated_files_are_ignored(tmp_path):
    """A stray best.pt / another run's prefix must not abort a periodic save (the reference's Saver ignores unrelated
    files, main.py:356-427); a file that carries OUR prefix without a step count is not one of ours either (the
    reference only parses the step token, main.py:401-411)."""
    import pytest
    d = str(tmp_path / "ckpt")
    net, app = _Net(), _Applier()
    app._create_slots(net.params.flat)
    os.makedirs(d)
    for stray in ("best.pt", "exported-model.pt", "other-12-300.pt"):
        open(os.path.join(d, stray), "wb").write(b"x")
    with pytest.warns(UserWarning):
        ck.save(d, net, app, global_t=100, wall_t=1.0)
    with pytest.warns(UserWarning):
        assert [t for t, _ in ck.list_checkpoints(d)] == [100]
        assert ck.restore(d, _Net(), _Applier())[0] == 100
    ck.save(d, net, app, global_t=300, wall_t=2.0, name="other")           # 'other-12-300.pt' is that run's own family
    open(os.path.join(d, "checkpoint-final.pt"), "wb").write(b"x")
    with pytest.warns(UserWarning):
        assert [t for t, _ in ck.list_checkpoints(d)] == [100]


def test_any_score_survives_save_restore_save(tmp_path):
    """str(abs(score))[2:8] is not always digits ('.5', '0.0', '.0', '-05'); the reference never parses the
    score field (main.py:401-411), so save -> restore -> save must work for every score, and files of other run families
    in the directory are ignored, not fatal."""
    import warnings
    for k, score in enumerate((-14.0, 12.5, 100.0, 1e-5, -1078.5, 0.0)):
        d = str(tmp_path / ("ckpt%d" % k))
        net, app = _Net(), _Applier()
        app._create_slots(net.params.flat)
        p1 = ck.save(d, net, app, global_t=100, wall_t=1.0, best_score=score)
        assert os.path.exists(p1)
        torch.save({"x": 1}, os.path.join(d, "checkpoint-best-final.pt"))      # another run family: same prefix, no step
        torch.save({"x": 1}, os.path.join(d, "best.pt"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net2, app2 = _Net(), _Applier()
            g, w, s = ck.restore(d, net2, app2)
            assert (g, w, s) == (100, 1.0, float(score))
            net.params.flat.add_(1.0)
            ck.save(d, net, app, global_t=300, wall_t=2.0, best_score=score)
            assert [t for t, _ in ck.list_checkpoints(d)] == [100, 300]
            assert ck.restore(d, net2, app2)[0] == 300 and torch.equal(net2.params.flat, net.params.flat)
