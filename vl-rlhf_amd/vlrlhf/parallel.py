"""Data-parallel gradient reduction for one-process-per-GPU DPO (the only mode the path keeps: accelerate_config/ddp.yaml,
MULTI_GPU).  xGMI is a point-to-point full mesh, so the flat bf16 gradient is reduced in a few LARGE contiguous buckets
(one decoder layer = ~0.4 GB at 7B) instead of DDP's 25 MB ones, each issued on a dedicated communication stream the moment
the HIP backward has finished writing it (reverse layer order) and overlapped with the remaining backward.  SUM
all-reduce; the 1/world_size is folded into the optimizer's gradient scale, so no extra pass over the gradients.

Two transports for the same buckets:
  * "native"  - vlr_allreduce_bucket of libvlr_hip.so (include/vlr.h): RCCL called straight from the C ABI on our comm
                stream; the unique id travels through torch.distributed once at start-up, and the communicator is verified
                with a known all-reduce before it is trusted.  The default.
  * "torch"   - torch.distributed all_reduce (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).  VLR_COMM=torch.
A native transport that cannot be initialised RAISES on every rank (the stages of NativeComm agree on success / failure among the
ranks, so nobody is left inside a collective): a multi-GPU run whose transport is not the one that was asked for should not silently
produce a number.  bench.py catches that error, says so in its JSON line and measures on the "torch" transport instead.

CUs for RCCL: the ring kernels of RCCL run one workgroup per CHANNEL beside the backward.  The persistent GEMM / attention launches leave
`comm_cus` CUs free (vlr_set_comm_cus; VLR_COMM_CUS, default 16 = two per XCD) and RCCL is bounded to that many channels so that the ring
kernels fit the reservation instead of displacing persistent workgroups (13 % of the step in the single-GPU interference bench,
profiles/r03_comm_cus_interference_1gpu.txt): our own communicator PER COMMUNICATOR (vlr_comm_init_cfg: ncclConfig_t maxCTAs / minCTAs,
ABI v8), torch's - only when it carries the buckets, VLR_COMM=torch - through the process-wide NCCL_MAX_NCHANNELS (rccl_channel_env(), set
before the first communicator exists; only the maximum is forced - a floor would take CUs the reservation does not cover)."""
import ctypes as C
import os
import subprocess
import sys
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def comm_cus_default() -> int:
    return int(os.environ.get("VLR_COMM_CUS", "16"))


def rccl_channel_env(env=None, comm_cus=None):
    """NCCL_MAX_NCHANNELS := comm_cus (unless the user set it) in `env` (default os.environ) - the bound on every communicator of the
    process that is not created through vlr_comm_init_cfg (torch.distributed's).  Must run before the first RCCL communicator of the
    process is created.  Returns the (max, min) strings in effect (min only when the user exported one)."""
    e = os.environ if env is None else env
    k = comm_cus_default() if comm_cus is None else comm_cus
    if k > 0:
        e.setdefault("NCCL_MAX_NCHANNELS", str(k))
    return e.get("NCCL_MAX_NCHANNELS"), e.get("NCCL_MIN_NCHANNELS")


def _all_ok(ok: bool, group=None) -> bool:
    """True iff `ok` on EVERY rank (one tiny all-reduce on the torch process group)"""
    flag = torch.tensor([1 if ok else 0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag))


class NativeComm:
    """RCCL communicator owned by libvlr_hip.so (one per process).  Construction is collective and staged so that a failure on one
    rank can never leave the others blocked inside ncclCommInitRank: (1) every rank loads RCCL through the library and the ranks
    agree that all could; (2) rank 0 creates the unique id and its success travels with the id; only then (3) all ranks join."""

    def __init__(self, group=None, channels: Optional[int] = None):
        """channels: bound of this communicator's RCCL channels (= ring-kernel workgroups); None = comm_cus_default(), 0 = unbounded"""
        from . import _hip
        self._hip = _hip
        self.comm = None
        self.channels = comm_cus_default() if channels is None else int(channels)
        self.channel_bound = "none"           # how the bound reached RCCL: "config" (ncclConfig_t maxCTAs) | "env" (NCCL_MAX_NCHANNELS) | "none"
        l = _hip.lib()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        # (1) the RCCL library resolves on every rank
        self.library = l.vlr_comm_library().decode()
        err = "" if self.library else l.vlr_last_error().decode()
        if not _all_ok(bool(self.library), group):
            raise _hip.VlrError(f"RCCL could not be loaded on every rank ({err or 'another rank failed'})")
        # (2) unique id from rank 0: [ok flag | id bytes]
        nbytes = _hip.helper("vlr_comm_unique_id_bytes")
        idt = torch.zeros(1 + nbytes, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            buf = (C.c_ubyte * nbytes)()
            if l.vlr_comm_unique_id(buf) == 0:
                idt[0] = 1
                idt[1:].copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
            else:
                err = l.vlr_last_error().decode()
        dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        host_id = idt.cpu()
        if int(host_id[0]) != 1:
            raise _hip.VlrError(f"rank 0 could not create an RCCL unique id ({err or 'see rank 0'})")
        # (3) join - with the channel bound in the communicator's own configuration; a library that does not take it (no
        # ncclCommInitRankConfig / configuration rejected: an argument error, raised before any rank talks to another) makes ALL ranks
        # fall back to the plain call under the process-wide NCCL_MAX_NCHANNELS
        host = (C.c_ubyte * nbytes).from_buffer_copy(bytes(host_id[1:].numpy().tobytes()))
        comm = C.c_void_p()
        rc, err = 1, ""
        #  whether the configured call can be made at all is agreed on BEFORE any rank enters it: the entry point exists and every
        # rank loaded the same RCCL version - a rank that failed there alone would leave the others waiting inside the bootstrap
        want_cfg = self.channels > 0 and os.environ.get("VLR_COMM_CONFIG", "1") != "0"
        if want_cfg:
            ver = torch.tensor([_hip.helper("vlr_comm_rccl_version")], dtype=torch.int64, device=dev)
            lo, hi = ver.clone(), ver.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            same = int(lo) == int(hi)
            if not _all_ok(bool(_hip.helper("vlr_comm_has_config")) and same, group):
                want_cfg = False
                self.config_error = "ncclCommInitRankConfig missing on a rank" if same else f"RCCL versions differ across the ranks ({int(lo)} .. {int(hi)})"
        if want_cfg:
            rc = l.vlr_comm_init_cfg(host, self.rank, self.world, 0, self.channels, C.byref(comm))
            err = "" if rc == 0 else l.vlr_last_error().decode()
            if _all_ok(rc == 0, group):
                self.channel_bound = "config"
            else:
                if rc == 0:
                    l.vlr_comm_destroy(comm)
                    comm = C.c_void_p()
                rc = 1
                # a new id: the first one may have been consumed by the ranks whose call went through
                idt.zero_()
                if self.rank == 0:
                    buf = (C.c_ubyte * nbytes)()
                    if l.vlr_comm_unique_id(buf) == 0:
                        idt[0] = 1
                        idt[1:].copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
                dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                host_id = idt.cpu()
                if int(host_id[0]) != 1:
                    raise _hip.VlrError("rank 0 could not create a second RCCL unique id")
                host = (C.c_ubyte * nbytes).from_buffer_copy(bytes(host_id[1:].numpy().tobytes()))
                self.config_error = err or "on another rank"
        if rc != 0:
            if self.channels > 0:
                self.channel_bound = "env" if os.environ.get("NCCL_MAX_NCHANNELS") == str(self.channels) else "none"
            rc = l.vlr_comm_init(host, self.rank, self.world, C.byref(comm))
            err = "" if rc == 0 else l.vlr_last_error().decode()
        if rc == 0:
            self.comm = comm
        if not _all_ok(rc == 0, group):
            self.close()
            raise _hip.VlrError(f"vlr_comm_init failed ({err or 'on another rank'})")
        # trust, but verify: sum of (rank + 1) over the ranks, in both dtypes the reducer uses
        good = True
        for dt in (torch.bfloat16, torch.float32):
            t = torch.full((1024,), float(self.rank + 1), dtype=dt, device=dev)
            self.all_reduce_(t, torch.cuda.current_stream())
            torch.cuda.current_stream().synchronize()
            want = self.world * (self.world + 1) / 2
            good = good and bool((t.float() == want).all())
        if not _all_ok(good, group):
            self.close()
            raise _hip.VlrError("vlr_allreduce_bucket self-check failed (sum of rank + 1 over the ranks)")

    def all_reduce_(self, t: torch.Tensor, stream):
        code = {torch.bfloat16: 0, torch.float32: 1}[t.dtype]
        l = self._hip.lib()
        if l.vlr_allreduce_bucket(self.comm, t.data_ptr(), t.numel(), code, stream.cuda_stream) != 0:
            raise self._hip.VlrError(l.vlr_last_error().decode())

    def close(self):
        if self.comm:
            self._hip.lib().vlr_comm_destroy(self.comm)
            self.comm = None


def make_transport(group=None, cuda=True):
    """-> (NativeComm | None, name, note).  CPU tensors (gloo tests) always use torch.distributed."""
    want = os.environ.get("VLR_COMM", "auto").lower()
    if not cuda or not dist.is_initialized() or dist.get_world_size(group) == 1 or want == "torch":
        return None, "torch", ""
    if dist.get_backend(group) != "nccl":       # e.g. gloo ranks sharing one GPU in the tests: RCCL refuses duplicate devices
        return None, "torch", f"process group backend is {dist.get_backend(group)}"
    # NativeComm's stages are collective and agree on success / failure among themselves: every rank either returns a working
    # communicator or raises - and a failure is FATAL here (VLR_COMM=torch runs on torch.distributed's RCCL communicator instead)
    comm = NativeComm(group)
    return comm, "native", comm.library


class GradReducer:
    def __init__(self, flat_grads: torch.Tensor, buckets: Dict[str, Tuple[int, int]], group=None, max_bucket_elems: int = 1 << 28):
        self.grads = flat_grads
        self.buckets = dict(buckets)
        self.group = group
        self.cuda = flat_grads.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.max_elems = max_bucket_elems
        self.enabled = True           # set False on non-final gradient-accumulation micro-steps (DDP no_sync)
        self._pending = []
        self._issued = False
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.native, self.transport, self.transport_note = make_transport(group, self.cuda)
        # CUs left to the RCCL ring kernels that run beside the backward: the persistent GEMM / attention launches size their grids to
        # (CUs - comm_cus).  Default 16 under data parallelism (2 per XCD), from the single-GPU interference bench (tools/
        # comm_interference.py); VLR_COMM_CUS overrides, 0 switches it off.  The CUs are given up only WHILE BUCKETS
        # ARE IN FLIGHT - from the first bucket of a backward to wait() - so the two forward passes (a third of the step, no exchange
        # beside them) and the whole of a LoRA step (one bucket, issued after the backward) keep the full chip; both forward passes of a
        # step see the same grid, which the policy == reference => loss == ln 2 identity needs.  VLR_COMM_CUS_SCOPE=step: the whole step
        # (rounds 3-5a).
        self.comm_cus = 0
        self.reserve_scope = os.environ.get("VLR_COMM_CUS_SCOPE", "backward").lower()
        self._reserved = False
        self.rccl_channels = (os.environ.get("NCCL_MAX_NCHANNELS"), os.environ.get("NCCL_MIN_NCHANNELS"))      # the process-wide bound (torch's communicator)
        self.channel_bound = self.native.channel_bound if self.native is not None else ("env" if os.environ.get("NCCL_MAX_NCHANNELS") else "none")
        if self.cuda and self.world > 1:
            self.comm_cus = comm_cus_default()
            if self.reserve_scope == "step":
                self._reserve(True)

    def _reserve(self, on: bool):
        """give `comm_cus` CUs up to / take them back from the ring kernels: the grids of the launches that FOLLOW on the host"""
        if self.cuda and self.comm_cus and on != self._reserved and (on or self.reserve_scope != "step"):
            from . import _hip
            _hip.helper("vlr_set_comm_cus", self.comm_cus if on else 0)
            self._reserved = on

    def bucket_ready(self, name: str):
        if not self.enabled or self.world == 1:
            return
        lo, hi = self.buckets[name]
        if hi <= lo:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            self._issued = True
            self._reserve(True)
            if self.native is not None:
                for a in range(lo, hi, self.max_elems):
                    self.native.all_reduce_(self.grads[a:min(hi, a + self.max_elems)], self.stream)
                return
            with torch.cuda.stream(self.stream):
                for a in range(lo, hi, self.max_elems):
                    self._pending.append(dist.all_reduce(self.grads[a:min(hi, a + self.max_elems)], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
        else:
            dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def reduce_all(self):
        """reduce every bucket now (used when the backward did not signal buckets, e.g. tests)."""
        for name in self.buckets:
            self.bucket_ready(name)
        self.wait()

    def wait(self):
        if self.cuda and self._issued:
            for w in self._pending:
                w.wait()
            torch.cuda.current_stream().wait_stream(self.stream)
            self._reserve(False)
        self._pending = []
        self._issued = False


def all_reduce_mean_scalars(values, device=None, group=None):
    """ONE all-reduce for the <= 9 logged scalars (loss + the eight DPO metrics): HF Trainer reports the loss averaged
    over the data-parallel ranks (`_nested_gather(tr_loss).mean()`); the metrics get the same treatment.  `values` is a
    list of floats / 0-d tensors; returns a list of floats.  No-op (local values) without a process group."""
    vals = [v.detach().float().reshape(()) if isinstance(v, torch.Tensor) else torch.tensor(float(v)) for v in values]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(v) for v in vals]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.stack([v.to(device) for v in vals])
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return (t / dist.get_world_size(group)).tolist()


def init_distributed_from_env(backend: Optional[str] = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torchrun / accelerate launch)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 0, 1
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (a launcher that did not export it: read when the HSA runtime starts, i.e. below)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        if os.environ.get("VLR_COMM", "auto").lower() == "torch":
            rccl_channel_env()           # the buckets travel on torch's communicator: bound it, before it is created.  (Native transport: our
                                         # communicator carries its bound in its own configuration and torch's only sees barriers and scalars.)
    if not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(script: str, argv, nproc: int, env=None) -> int:
    """`python script --gpus N` without a launcher: re-run it as N ranks of one node (one per GPU) under
    torch.distributed.run on 127.0.0.1 - what accelerate launch does for the reference (accelerate_config/ddp.yaml:
    MULTI_GPU, num_machines 1, num_processes 8).  Returns the launcher's exit code; rank 0's stdout passes through."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, nproc))))
    if nproc > 1 and e.get("VLR_COMM", "auto").lower() == "torch":
        rccl_channel_env(e)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=e)
