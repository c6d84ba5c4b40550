"""rr_allreduce_f32: the gradient exchange of a data-parallel job behind the C-ABI.  One GPU per
box here, so the communicator has ONE rank: what is checked is the binding (RCCL resolved at run time, communicator made
through rr_comm_unique_id / rr_comm_init_rank, the collective enqueued on the caller's stream, the scaling pass, status
codes) - the N-rank arithmetic is RCCL's own; the weighting that makes the reduced gradient equal the single-process one
is proven with 2 gloo processes in tests/test_dp_gloo.py."""
import ctypes as C

import pytest
import torch

from reactranker_amd._lib import check, lib, ptr, stream


def test_allreduce_argument_checks_need_no_gpu():
    l = lib()
    assert l.rr_allreduce_f32(None, 0, 1.0, None, None) == -1            # RR_ERR_ARG: null buffer / communicator
    assert l.rr_allreduce_rsag_f32(None, 0, 1.0, None, None) == -1
    assert l.rr_comm_destroy(None) == -1
    assert l.rr_comm_init_rank(None, 1, None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n,scale", [(790_101, 0.5), (64, 1.0), (3, 0.25)])
def test_allreduce_over_a_one_rank_communicator(n, scale):
    l = lib()
    ident = (C.c_char * 128)()
    check(l.rr_comm_unique_id(ident), "rr_comm_unique_id")
    comm = C.c_void_p()
    check(l.rr_comm_init_rank(C.byref(comm), 1, ident, 0), "rr_comm_init_rank")
    try:
        x = torch.randn(n + 1, device="cuda")[1:]                           # (an unaligned start exercises the scalar tail)
        y = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                        # enqueued on the caller's stream, not the null stream
            check(l.rr_allreduce_f32(ptr(y), n, scale, comm, stream()), "rr_allreduce_f32")
        side.synchronize()
        assert torch.equal(y, x * scale)
        check(l.rr_allreduce_f32(ptr(y), 0, 1.0, comm, stream()), "empty buffer")
        # the two-collective form (reduce-scatter + all-gather over the prefix that divides by the rank count, ABI revision 8):
        # same result on the same communicator
        z = x.clone()
        check(l.rr_allreduce_rsag_f32(ptr(z), n, scale, comm, stream()), "rr_allreduce_rsag_f32")
        torch.cuda.synchronize()
        assert torch.equal(z, y)
    finally:
        check(l.rr_comm_destroy(comm), "rr_comm_destroy")


@pytest.mark.gpu
def test_rccl_already_in_the_process_is_used_as_is():
    """A host that links RCCL has ncclAllReduce among its symbols: rr_comm_* must bind to THAT copy (dlsym(RTLD_DEFAULT) -
    a null handle on glibc, which round 3's resolver mistook for "not found") instead of opening a second one.  A fresh
    interpreter with librccl preloaded stands in for such a host."""
    import glob
    import os
    import subprocess
    import sys
    cands = sorted(glob.glob("/opt/rocm*/lib/librccl.so.1")) + sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
    if not cands:
        pytest.skip("no librccl on this machine")
    code = ("import ctypes as C; from reactranker_amd._lib import lib; l = lib(); ident = (C.c_char * 128)(); "
            "rc = l.rr_comm_unique_id(ident); print('RC', rc, 'HOW', l.rr_comm_backend())")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_PRELOAD=cands[0], PYTHONPATH=repo)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "RC 0 HOW 1" in out.stdout, (out.stdout, out.stderr[-2000:])
    env.pop("LD_PRELOAD")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "RC 0 HOW" in out.stdout, (out.stdout, out.stderr[-2000:])          # nothing preloaded: torch's copy, or dlopen


@pytest.mark.gpu
def test_gradient_bucket_rsag_over_a_one_rank_rccl_group(tmp_path):
    """dp.GradBucket with algo = "rsag" over torch.distributed's nccl (= RCCL) backend: the reduce-scatter + all-gather path runs
    on a real communicator (one rank on a one-GPU box: the sum of one is itself) and leaves the gradients where autograd
    expects them.  A fresh process, because a process group can be initialised only once."""
    import os
    import subprocess
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from reactranker_amd import dp\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "torch.cuda.set_device(0)\n"
        "m = torch.nn.Linear(37, 5).cuda()\n"
        "b = dp.GradBucket(m.parameters())\n"
        "m(torch.randn(3, 37, device='cuda')).sum().backward()\n"
        "ref = [p.grad.clone() for p in m.parameters()]\n"
        "dp.GradBucket.algo = 'rsag'\n"
        "b.flat.copy_(torch.cat([g.reshape(-1) for g in ref]))\n"
        "b._collective(None)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(b.flat, torch.cat([g.reshape(-1) for g in ref]))\n"
        "print('RSAG OK', dist.get_backend())\n"
        "dist.destroy_process_group()\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "RSAG OK nccl" in out.stdout, (out.stdout, out.stderr[-3000:])
