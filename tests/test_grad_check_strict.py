"""run_grad_check as a parity gate, without relaxed thresholds.

The reference's GradChecker (src/grad_check.cc) compares the analytic gradient with central differences of an fp32 loss.  In fp32
that check is noisy for the reference itself: on its own CPU path it fails some weight checks of every net here (finite-difference
round-off ~ ulp(L) / (2 eps batch) against ReLU / max-pool kinks at larger eps; an all-zero gradient is 0/0 = NaN = FAILED,
grad_check.cc:59-61).  So the gate is stated relative to the reference, at the SAME point:

  * identical parameters (ref_host.golden_params: integer hash, He scale) and identical batch (the data shim's hash batch 0) are
    given to the reference's compiled GradChecker on the reference's CPU path, to the same compiled GradChecker on this library
    (oracle/_ref/libref_host_{cpu,hip}.so, seam_host.cc SeamGradChecker::RunFixed) and to this repo's port;
  * every check the reference-CPU run PASSES (its own compiled verdict, grad_check.cc:61) must PASS on this library — except that a
    check the CPU run passes by a hair (its criterion value within a factor 2 of the 1 % limit) may fail here by a hair (within a
    factor 2 on the other side): the quantity is noisy at that level on any fp32 machine, and at most one such case per net is accepted;
  * every analytic gradient (all checks, passed or not) agrees with the CPU run to 1e-4;
  * the python port reaches the library-side verdicts of the reference's own checker and the same analytic numbers.
"""
import os

import numpy as np
import pytest

import ref_host
from golden_cases import rel_err

SEED = 5


def _text(which):
    from convnet_amd import models
    from test_net_gpu import small_alexnet
    return {"tiny_alex": lambda: small_alexnet(grad_check=True), "lenet5": lambda: models.lenet5(grad_check=True),
            "mnist_conv": lambda: models.mnist_conv(grad_check=True)}[which]()


def _reference_run(host, tag, which, batch, tmp):
    m, d = ref_host.write_configs(tmp, _text(which), batch, 1, SEED, which)
    layers, edges, total = host.describe(m, d)
    slices, end = ref_host.slices_from_describe(layers, edges)
    assert end == total
    p0 = ref_host.golden_params(total, SEED, slices)
    out = os.path.join(str(tmp), f"gc_{which}_{tag}.h5")
    flags = host.grad_check_fixed(m, d, p0, out)
    names = [f"{s}:{t}" for s, t, n in edges if n]
    assert len(flags) == len(names)
    return names, flags, ref_host.read_grad_check(out, names), p0


@pytest.mark.parametrize("which,batch", [("lenet5", 16), ("tiny_alex", 8)])
def test_reference_cpu_grad_checker_at_the_fixed_point(which, batch, tmp_path):
    """CPU only: the reference's checker on its own CPU path passes some checks and fails others at this point — the measured
    form of 'fp32 grad_check is noisy for the reference too' — and the python restatement of its rule reproduces its verdicts
    where no epsilon carry-over is involved (first epsilon)."""
    if not os.path.exists(ref_host.CPU_SO):
        pytest.skip("oracle/_ref/libref_host_cpu.so not built")
    names, flags, res, _ = _reference_run(ref_host.RefHost(ref_host.CPU_SO), "cpu", which, batch, tmp_path)
    verdicts = [v for pair in flags for v in pair]
    assert any(verdicts) and len(verdicts) == 2 * len(names)
    for name, (fw, fb) in zip(names, flags):
        for kind, passed in (("weights", fw), ("bias", fb)):
            a, n = res[name][kind]
            if not np.any(a) and not np.any(n):
                assert not passed, "all-zero gradient: 0/0 -> FAILED in the reference (grad_check.cc:59-61)"
            elif ref_host.grad_check_passes(a, n[:1])[0]:
                assert passed, (name, kind)
            # the restatement of the running criterion (carry-over included) reproduces the compiled verdict, check by check
            crit = ref_host.grad_check_criterion(a, n)
            assert (crit[-1] < 0.01) == passed, (name, kind, crit, passed)


@pytest.mark.gpu
@pytest.mark.parametrize("which,batch", [("lenet5", 16), ("tiny_alex", 8), ("mnist_conv", 8)])
def test_this_library_passes_every_check_the_reference_cpu_passes(which, batch, tmp_path):
    import ctypes
    import torch
    assert torch.cuda.is_available()
    from convnet_amd import _lib
    from convnet_amd.grad_check import GradChecker
    from convnet_amd.matrix import Matrix
    from test_reference_host import HashDataHandler
    # the parity GATE of the GPU suite: on a GPU box a missing checker is a failure, not a skip (oracle/_ref is built here by
    # __graft_entry__.build() from /root/reference and travels with the snapshot)
    assert os.path.exists(ref_host.CPU_SO) and os.path.exists(ref_host.HIP_SO), \
        "oracle/_ref/libref_host_{cpu,hip}.so missing: run `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference exists"
    Matrix.SetupCUDADevice(0)
    ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    names, cpu_flags, cpu_res, p0 = _reference_run(ref_host.RefHost(ref_host.CPU_SO), "cpu", which, batch, tmp_path)
    names2, hip_flags, hip_res, _ = _reference_run(ref_host.RefHost(ref_host.HIP_SO), "hip", which, batch, tmp_path)
    assert names == names2
    # this repo's port of the checker, same point
    net = GradChecker(_text(which), fused=False)
    net.SetBatchsize(batch)
    net.SetupDataset(HashDataHandler(net, batch, 1, SEED))
    net.AllocateMemory(False)
    net.parameters_.FromNumpy(p0)
    port = net.Run(fixed_batch=True)
    n_cpu = n_hip = 0
    report, borderline = [], []
    for name, (cw, cb), (hw, hb) in zip(names, cpu_flags, hip_flags):
        for kind, c, h in (("weights", cw, hw), ("bias", cb, hb)):
            a_cpu, num_cpu = cpu_res[name][kind]
            a_hip, num_hip = hip_res[name][kind]
            n_cpu += c
            n_hip += h
            report.append((name, kind, c, h, ref_host.grad_check_passes(a_cpu, num_cpu)[1], ref_host.grad_check_passes(a_hip, num_hip)[1]))
            # analytic gradient: same numbers on both machines, passed or not
            if np.any(a_cpu) or np.any(a_hip):
                assert rel_err(a_hip, a_cpu) < 1e-4, ("analytic", name, kind, a_hip, a_cpu)
            else:
                assert np.array_equal(a_hip, a_cpu)
            # the gate: whatever the reference's CPU run passes, this library passes (the reference's compiled verdict both times).
            # The verdict is a threshold on a noisy quantity (finite differences of an fp32 loss: the two machines sum in different
            # orders, so the numerical gradients differ at the 1e-4 level), so a check the CPU run passes BY A HAIR can land on the
            # other side here.  "By a hair" is made explicit instead of tolerated silently: the CPU run's passing value of the
            # reference's own criterion must be above half the 1 % limit, and this library's best value must stay under twice it.
            crit_cpu, crit_hip = ref_host.grad_check_criterion(a_cpu, num_cpu), ref_host.grad_check_criterion(a_hip, num_hip)
            assert (crit_cpu[-1] < 0.01) == c and (crit_hip[-1] < 0.01) == h, ("python restatement of the criterion != compiled verdict", name, kind)
            if c and not h:
                assert crit_cpu[-1] >= 0.005 and np.nanmin(crit_hip) < 0.02, ("reference CPU passes clearly, this library fails", name, kind,
                                                                                crit_cpu, crit_hip, a_cpu, a_hip)
                borderline.append((name, kind, crit_cpu[-1], float(np.nanmin(crit_hip))))
            # the port: same verdict and the same analytic numbers as the reference's checker on the same library
            p_pass, p_a, _ = port[name][kind]
            assert bool(p_pass) == h, ("port verdict", name, kind, p_pass, h)
            assert (rel_err(p_a, a_hip) < 1e-5) if np.any(a_hip) else (not np.any(p_a)), ("port analytic", name, kind)
    print(f"{which}: reference CPU passes {n_cpu}, on this library {n_hip} of {2 * len(names)} checks")
    for r in report:
        print("   ", r)
    print("    borderline (CPU passes within a factor 2 of the limit, this library does not):", borderline)
    assert n_cpu > 0 and n_hip >= n_cpu - len(borderline) and len(borderline) <= 1
