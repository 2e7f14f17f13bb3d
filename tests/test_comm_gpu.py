"""vlr_comm_* / vlr_allreduce_bucket (include/vlr.h): RCCL reached from the C ABI.  One MI355X is all a test box has, and RCCL
refuses two ranks on one device, so this covers what a single rank can: the library resolves and loads RCCL, creates a
communicator from a unique id, and an in-place SUM all-reduce over a 1-rank world returns the data unchanged, on a side
stream, for both dtypes; argument errors come back as status codes.  The multi-rank wiring (id broadcast, self-check,
fallback) is exercised with world_size 2 on CPU/gloo in tests/test_ddp_gloo.py and by the driver's scaling bench."""
import ctypes as C

import pytest
import torch


def test_comm_entry_points_report_argument_errors_without_gpu():
    from vlrlhf import _hip
    l = _hip.lib()
    assert _hip.helper("vlr_comm_unique_id_bytes") == 128
    assert l.vlr_comm_unique_id(None) == 1 and b"null" in l.vlr_last_error()
    assert l.vlr_comm_init(None, 0, 1, None) == 1
    buf = (C.c_ubyte * 128)()
    comm = C.c_void_p()
    assert l.vlr_comm_init(buf, 3, 2, C.byref(comm)) == 1 and b"outside world" in l.vlr_last_error()
    assert l.vlr_allreduce_bucket(None, None, 8, 0, None) == 1 and b"communicator" in l.vlr_last_error()
    assert l.vlr_comm_destroy(None) == 0
    assert l.vlr_comm_init_cfg(None, 0, 1, 0, 16, None) == 1
    assert l.vlr_comm_init_cfg(buf, 0, 1, 20, 16, C.byref(comm)) == 1 and b"min_ctas" in l.vlr_last_error()


@pytest.mark.gpu
def test_single_rank_allreduce_through_the_c_abi():
    from vlrlhf import _hip
    l = _hip.lib()
    lib_path = l.vlr_comm_library().decode()
    assert "rccl" in lib_path, l.vlr_last_error()
    buf = (C.c_ubyte * 128)()
    assert l.vlr_comm_unique_id(buf) == 0, l.vlr_last_error()
    comm = C.c_void_p()
    assert l.vlr_comm_init(buf, 0, 1, C.byref(comm)) == 0, l.vlr_last_error()
    side = torch.cuda.Stream()
    for dt, code in ((torch.bfloat16, 0), (torch.float32, 1)):
        x = torch.randn(1 << 20, device="cuda").to(dt)
        want = x.clone()
        side.wait_stream(torch.cuda.current_stream())
        assert l.vlr_allreduce_bucket(comm, x.data_ptr(), x.numel(), code, side.cuda_stream) == 0, l.vlr_last_error()
        side.synchronize()
        assert torch.equal(x, want)
    assert l.vlr_allreduce_bucket(comm, x.data_ptr(), 8, 5, None) == 1 and b"dtype" in l.vlr_last_error()
    assert l.vlr_comm_destroy(comm) == 0
    print(f"[comm] RCCL library in use: {lib_path}")


@pytest.mark.gpu
def test_channel_bounded_communicator_through_the_c_abi():
    """vlr_comm_init_cfg (ABI v8): the communicator takes its channel bound (ncclConfig_t maxCTAs, NCCL 2.18 layout) from the call, not
    from the process environment - the RCCL loaded at run time (PyTorch's) accepts the configuration, and an all-reduce on it works"""
    from vlrlhf import _hip
    l = _hip.lib()
    ver = _hip.helper("vlr_comm_rccl_version")
    assert ver >= 21700, ver
    buf = (C.c_ubyte * 128)()
    assert l.vlr_comm_unique_id(buf) == 0, l.vlr_last_error()
    comm = C.c_void_p()
    assert l.vlr_comm_init_cfg(buf, 0, 1, 0, 16, C.byref(comm)) == 0, l.vlr_last_error()
    x = torch.randn(1 << 20, device="cuda").bfloat16()
    want = x.clone()
    assert l.vlr_allreduce_bucket(comm, x.data_ptr(), x.numel(), 0, torch.cuda.current_stream().cuda_stream) == 0, l.vlr_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    assert l.vlr_comm_destroy(comm) == 0
    print(f"[comm] RCCL version {ver}: ncclCommInitRankConfig(maxCTAs = 16) accepted")


@pytest.mark.gpu
def test_native_comm_staged_construction_one_rank_nccl():
    """parallel.NativeComm on a real RCCL process group of one rank: the staged, collective construction (library check ->
    unique id with its ok flag -> join -> self-check) and an in-place bucket reduction on a side stream."""
    import torch.distributed as dist
    from vlrlhf.parallel import NativeComm, free_port
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    try:
        nc = NativeComm()
        assert "rccl" in nc.library and nc.world == 1
        assert nc.channel_bound == "config" and nc.channels == 16, (nc.channel_bound, getattr(nc, "config_error", ""))
        g = torch.randn(1 << 22, device="cuda").bfloat16()
        want = g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        nc.all_reduce_(g[1000:1 << 21], side)
        side.synchronize()
        assert torch.equal(g, want)
        nc.close()
        assert nc.comm is None
    finally:
        if created:
            dist.destroy_process_group()
