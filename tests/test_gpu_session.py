"""Resident sessions (dc_hip_session_*, -m gpu): the fused pop -> FE -> NN -> sigma2 -> forest flow on
coordinates uploaded once must equal the call-by-call path bit for bit.  The multi-device code path (one
host thread per device, one segment each, partials merged) runs
  - with two REAL devices over RCCL wherever the box has two (the tests then REQUIRE two devices and RCCL:
    no silent degradation to one rank),
  - on a one-GPU box through a one-rank RCCL communicator (DC_SESSION_FORCE_RCCL=1), and
  - on ANY box as two / three "devices" that share the one GPU (DC_SESSION_ALLOW_DUPLICATE_DEVICES=1: own
    host thread, stream, buffers and segment each) merged through the host like the reference does
    (density_clustering_cuda.cu:171-180, :311-326) -- the fallback of a session whose RCCL is missing."""
import os
import subprocess
import sys

import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dens():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from clustering_amd import density
    return density


def call_by_call(dens, c, radii, fe_index):
    import torch
    ct = torch.from_numpy(c).cuda()
    pops = dens.calculate_populations_partial(ct, radii)
    fe = dens.calculate_free_energies(pops[fe_index].contiguous())
    nn = dens.nearest_neighbors_partial(ct, fe)
    return (pops.cpu().numpy().astype(np.uint32), fe.cpu().numpy(),
            [t.cpu().numpy() for t in nn], dens.compute_sigma2(nn[1]))


def check_session(dens, s, c, radii, fe_index):
    want_p, want_fe, want_nn, want_s2 = call_by_call(dens, c, radii, fe_index)
    pops = s.populations(radii)
    assert (pops == want_p).all()
    fe = s.free_energies(fe_index)
    assert (bits(fe) == bits(want_fe)).all()
    nn_idx, nn_d2, hd_idx, hd_d2, s2 = s.nearest_neighbors()
    assert (nn_idx == want_nn[0].astype(np.uint32)).all() and (hd_idx == want_nn[2].astype(np.uint32)).all()
    assert (bits(nn_d2) == bits(want_nn[1])).all() and (bits(hd_d2) == bits(want_nn[3])).all()
    assert s2 == want_s2
    return fe, nn_d2, s2


@pytest.mark.parametrize("n_rows,n_cols", [(30000, 10), (5000, 30), (257, 3), (2000, 70)])
def test_session_equals_the_call_by_call_path(dens, n_rows, n_cols):
    c = gaussian_blobs(n_rows, n_cols, seed=3 + n_cols)
    radii = [0.25, 0.15, 0.4] if n_cols <= 10 else [0.7, 0.5]
    with dens.Session(c, n_devices=1) as s:
        assert s.n_devices == 1 and not s.uses_rccl
        fe, nn_d2, s2 = check_session(dens, s, c, radii, 1)
        # the second pass of the no-radius flow (density_clustering.cpp:649-673) on the same resident
        # coordinates: populations at the lumping radius, free energies, neighbours
        r_lump = float(np.float32(np.sqrt(4.0 * s2)))
        check_session(dens, s, c, [r_lump], 0)
        # free energies handed in by the caller (-D re-use)
        s.set_free_energies(fe)
        again = s.nearest_neighbors()
        assert (bits(again[1]) == bits(nn_d2)).all()
        if n_cols <= 64:
            rank = np.argsort(np.argsort(fe, kind="stable"), kind="stable").astype(np.uint32)
            edges, rounds = s.radius_forest(4.0 * s2, rank)
            want, _ = dens.radius_forest(c, np.float32(4.0 * s2), rank)
            norm = lambda e: sorted((int(min(a, b)), int(max(a, b))) for a, b in e)
            assert norm(edges) == norm(want)


def test_density_all_on_two_devices_equals_one(dens):
    """dc_hip_density_all with n_devices = min(available, 2) against one device (on a one-GPU box both are
    the same single-device session; the two-device merge then runs in the next test's one-rank form)."""
    import ctypes as C
    from clustering_amd import capi
    n, d = 20000, 10
    c = gaussian_blobs(n, d, seed=9)
    radii = np.array([0.3, 0.2], dtype=np.float32)
    avail = capi.device_count()

    def run(n_dev):
        pops = np.zeros((2, n), dtype=np.uint32)
        fe = np.zeros(n, dtype=np.float32)
        nn_idx, hd_idx = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        nn_d2, hd_d2 = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(capi.lib.dc_hip_density_all(p(c), n, d, p(radii), 2, 1, n_dev, p(pops), p(fe), p(nn_idx), p(nn_d2),
                                               p(hd_idx), p(hd_d2)), "dc_hip_density_all")
        return pops, fe, nn_idx, nn_d2, hd_idx, hd_d2

    if avail >= 2:   # two devices exist: the two-device session must really be one (and merge over RCCL)
        with dens.Session(c[:256], n_devices=2) as s2:
            assert s2.n_devices == 2 and s2.merge_mode in (1, 2), "a 2-device session degraded to one rank"
            assert s2.uses_rccl, "two devices, but the session merges on the host: RCCL did not come up"
    one = run(1)
    two = run(min(avail, 2))
    for a, b in zip(one, two):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    want_p, want_fe, want_nn, _ = call_by_call(dens, c, [0.3, 0.2], 1)
    assert (one[0] == want_p).all() and (bits(one[1]) == bits(want_fe)).all()
    assert (one[2] == want_nn[0].astype(np.uint32)).all() and (bits(one[5]) == bits(want_nn[3])).all()


RCCL_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = gaussian_blobs(12000, 10, seed=21)
with dens.Session(c, n_devices=int(sys.argv[2])) as s:
    assert s.uses_rccl and s.merge_mode == 1, "the session did not build an RCCL communicator"
    assert s.n_devices == int(sys.argv[2]), "fewer devices than asked for"
    print("devices", s.n_devices)
    pops = s.populations([0.2, 0.3])
    fe = s.free_energies(0)
    nn = s.nearest_neighbors()
    rank = np.argsort(np.argsort(fe, kind="stable"), kind="stable").astype(np.uint32)
    edges, rounds = s.radius_forest(4.0 * nn[4], rank)
np.savez(sys.argv[3], pops=pops, fe=fe, nn_idx=nn[0], nn_d2=nn[1], hd_idx=nn[2], hd_d2=nn[3], sigma2=nn[4],
         edges=np.array(sorted((int(min(a, b)), int(max(a, b))) for a, b in edges)))
"""


@pytest.mark.parametrize("nn_merge", ["allgather", "allreduce"])
def test_session_over_rccl(dens, tmp_path, nn_merge):
    """The merge path of a multi-GPU session -- segment sweeps, ncclAllReduce(sum) of the populations, the
    neighbours as an ncclAllGather of position-ordered blocks (default) or an ncclAllReduce(min) of packed words
    (DC_SESSION_NN_MERGE=allreduce), all in group calls -- with every device the box has; on a one-GPU box as a
    one-rank communicator."""
    from clustering_amd import capi
    avail = capi.device_count()
    out = str(tmp_path / "r.npz")
    env = dict(os.environ, DC_SESSION_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", DC_SESSION_NN_MERGE=nn_merge)
    r = subprocess.run([sys.executable, "-c", RCCL_CHILD, ROOT, str(min(avail, 2)), out], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    c = gaussian_blobs(12000, 10, seed=21)
    want_p, want_fe, want_nn, want_s2 = call_by_call(dens, c, [0.2, 0.3], 0)
    assert (got["pops"] == want_p).all() and (bits(got["fe"]) == bits(want_fe)).all()
    assert (got["nn_idx"] == want_nn[0].astype(np.uint32)).all() and (got["hd_idx"] == want_nn[2].astype(np.uint32)).all()
    assert (bits(got["nn_d2"]) == bits(want_nn[1])).all() and (bits(got["hd_d2"]) == bits(want_nn[3])).all()
    assert float(got["sigma2"]) == want_s2
    rank = np.argsort(np.argsort(want_fe, kind="stable"), kind="stable").astype(np.uint32)
    want_e, _ = dens.radius_forest(c, np.float32(4.0 * want_s2), rank)
    assert got["edges"].tolist() == sorted([int(min(a, b)), int(max(a, b))] for a, b in want_e)


def test_bench_starts_its_own_ranks_and_the_sharded_step_agrees(tmp_path):
    """`bench.py --gpus N` without a launcher spawns its N ranks itself (torch.distributed.run as a child, before
    anything touches HIP).  On a one-GPU box the ranks share the device and talk over gloo
    (DC_BENCH_ONE_DEVICE=1: timings mean nothing, the sharded step -- segment sweeps, all-reduce(sum) of the
    populations, all-reduce(min) of the packed neighbour words -- is the real one): its check values must equal
    the single-rank run's."""
    import json
    env = dict(os.environ, DC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    args = ["--steps", "1", "--warmup", "1", "--n-rows", "150000", "--cpu-sample", "0"]
    out = {}
    for n in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args,
                           capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out[n] = json.loads(line)
        assert out[n]["n_gpus"] == n and out[n]["roofline"]["frac"] <= 1.0
    assert out[4]["config"]["rccl_ranks"] == 4 and out[4]["config"]["backend"] == "gloo"
    assert out[1]["check"] == out[4]["check"]        # mean / max population and sigma2 of the merged results


HOST_MERGE_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n_dev = int(sys.argv[2])
c = gaussian_blobs(12000, 10, seed=21)
torch.cuda.set_device(0)
with dens.Session(c, devices=[0] * n_dev) as s:
    assert s.n_devices == n_dev and s.merge_mode == 2 and not s.uses_rccl
    assert "THROUGH THE HOST" in s.merge_note and "more than once" in s.merge_note, s.merge_note
    pops = s.populations([0.2, 0.3])
    fe = s.free_energies(0)
    nn = s.nearest_neighbors()
    rank = np.argsort(np.argsort(fe, kind="stable"), kind="stable").astype(np.uint32)
    edges, rounds = s.radius_forest(4.0 * nn[4], rank)
    assert torch.cuda.current_device() == 0          # every entry point restores the caller's device
np.savez(sys.argv[3], pops=pops, fe=fe, nn_idx=nn[0], nn_d2=nn[1], hd_idx=nn[2], hd_d2=nn[3], sigma2=nn[4],
         edges=np.array(sorted((int(min(a, b)), int(max(a, b))) for a, b in edges)))
"""


@pytest.mark.parametrize("n_dev,nn_merge", [(2, "allgather"), (3, "allgather"), (3, "allreduce")])
def test_session_host_merge_of_several_segments(dens, tmp_path, n_dev, nn_merge):
    """The multi-device flow with its HOST merge (what a session falls back to when RCCL cannot be loaded or its
    communicator cannot be built; the reference's own merge, density_clustering_cuda.cu:171-180, :311-326): n_dev
    "devices" on the one physical GPU -- one host thread, stream, workspace and segment each, sweeping concurrently --
    partial populations summed, the neighbours merged as gathered position-ordered blocks (default: what the RCCL
    path does with ncclAllGather) or as minimised packed words (DC_SESSION_NN_MERGE=allreduce); results and forest
    equal to the single-device call-by-call path bit for bit."""
    out = str(tmp_path / "h.npz")
    env = dict(os.environ, DC_SESSION_ALLOW_DUPLICATE_DEVICES="1", DC_SESSION_NN_MERGE=nn_merge)
    r = subprocess.run([sys.executable, "-c", HOST_MERGE_CHILD, ROOT, str(n_dev), out], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    c = gaussian_blobs(12000, 10, seed=21)
    want_p, want_fe, want_nn, want_s2 = call_by_call(dens, c, [0.2, 0.3], 0)
    assert (got["pops"] == want_p).all() and (bits(got["fe"]) == bits(want_fe)).all()
    assert (got["nn_idx"] == want_nn[0].astype(np.uint32)).all() and (got["hd_idx"] == want_nn[2].astype(np.uint32)).all()
    assert (bits(got["nn_d2"]) == bits(want_nn[1])).all() and (bits(got["hd_d2"]) == bits(want_nn[3])).all()
    assert float(got["sigma2"]) == want_s2
    rank = np.argsort(np.argsort(want_fe, kind="stable"), kind="stable").astype(np.uint32)
    want_e, _ = dens.radius_forest(c, np.float32(4.0 * want_s2), rank)
    assert got["edges"].tolist() == sorted([int(min(a, b)), int(max(a, b))] for a, b in want_e)


def test_duplicate_devices_need_the_test_switch(dens):
    """without DC_SESSION_ALLOW_DUPLICATE_DEVICES a device listed twice is an argument error"""
    from clustering_amd import capi
    c = gaussian_blobs(256, 5, seed=1)
    if os.environ.get("DC_SESSION_ALLOW_DUPLICATE_DEVICES") == "1":
        pytest.skip("switch set in the environment")
    with pytest.raises(capi.DensityLibraryError):
        dens.Session(c, devices=[0, 0])


@pytest.mark.parametrize("env_list,n_devices,ok", [("0", 0, True), ("0", 1, True), ("0,x", 0, False), ("0,", 0, False),
                                                   ("-1", 0, False), ("0,0", 3, False), ("7,", 1, False)])
def test_session_device_list_from_the_environment_never_overrides_silently(dens, monkeypatch, env_list, n_devices, ok):
    """DC_SESSION_DEVICES (for hosts that do not choose devices themselves): a malformed list is an argument error, not a
    shorter list, and a caller that asked for N devices -- other than all that are present, which is what hosts pass for
    "all of them" -- gets the list only if it names N"""
    from clustering_amd import capi
    monkeypatch.setenv("DC_SESSION_DEVICES", env_list)
    monkeypatch.delenv("DC_SESSION_ALLOW_DUPLICATE_DEVICES", raising=False)
    c = gaussian_blobs(256, 5, seed=1)
    if ok:
        with dens.Session(c, n_devices=n_devices) as s:
            assert s.n_rows == 256
    else:
        with pytest.raises(capi.DensityLibraryError):
            dens.Session(c, n_devices=n_devices)
