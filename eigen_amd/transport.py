"""How a panel piece gets from the process that owns it to the others (one process per GPU).

Two device transports:

* ``ipc``  -- every receiver PULLS the piece out of the owner's HBM: the owner's buffer is opened once
  through an inter-process memory handle (``eigamd_ipc_export`` / ``eigamd_ipc_open``) and a transfer is
  one device-to-device copy on the receiver's copy stream.  The DMA engines move it over the xGMI link
  of that pair of devices, so a grid row's receivers drive as many links at once and no compute unit is
  taken from the local product.  Ranks may share a device (the tests put 2 and 4 on one GPU).
* ``rccl`` -- ``torch.distributed.broadcast`` with backend "nccl" (RCCL).  Its kernels need compute units
  while the local product holds every one of them with a workgroup that runs for a whole tile, so the
  local products are launched on a stream that leaves a few compute units alone
  (``eigamd_stream_create_cu_mask``).

The reference has nothing of the kind: its threads share one packed panel in memory
(Eigen/src/Core/products/GeneralMatrixMatrix.h:100-136); this module is that sharing between processes.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import lib


def _export(tensor):
    h = ctypes.create_string_buffer(_lib.IPC_HANDLE_BYTES)
    off, span = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.eigamd_ipc_export(ctypes.c_void_p(tensor.data_ptr()), h, ctypes.byref(off), ctypes.byref(span)),
               "eigamd_ipc_export")
    return h.raw, off.value


class PeerMemory:
    """Allocations of other processes opened in this one; one entry per (process, allocation), however many
    tensors live in that allocation (torch's allocator carves many tensors out of one)."""

    def __init__(self):
        self._base = {}

    def address(self, pid, handle, offset):
        key = (pid, handle)
        if key not in self._base:
            base = ctypes.c_void_p()
            _lib.check(lib.eigamd_ipc_open(handle, ctypes.byref(base)), "eigamd_ipc_open")
            self._base[key] = base.value
        return self._base[key] + offset

    def close(self):
        for base in self._base.values():
            lib.eigamd_ipc_close(ctypes.c_void_p(base))
        self._base.clear()


class Window:
    """The tensors `tensor` of all `ranks` (each rank passes its own), readable from this process:
    ``ptr[rank]`` is an address valid here.  Collective over `group` (the host-side group)."""

    def __init__(self, tensor, ranks, group, peers):
        me = dist.get_rank()
        handle, offset = _export(tensor)
        everyone = [None] * len(ranks)
        dist.all_gather_object(everyone, (os.getpid(), handle, offset), group=group)
        self.tensor = tensor   # (kept alive: peers read it)
        self.ptr = {}
        for rk, (pid, h, off) in zip(ranks, everyone):
            self.ptr[rk] = tensor.data_ptr() if rk == me else peers.address(pid, h, off)


def copy_async(dst_ptr, src_ptr, nbytes, stream):
    _lib.check(lib.eigamd_copy_dev_async(ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), int(nbytes),
                                         ctypes.c_void_p(stream.cuda_stream)), "eigamd_copy_dev_async")


def host_group():
    """A process group whose collectives are host-side only (barriers and small objects between the ranks'
    host threads must not queue anything on a device): the default group when it is gloo, else a gloo group
    over the same ranks.  Collective on first use."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    g = getattr(host_group, "_group", None)
    if g is None:
        g = host_group._group = dist.new_group(backend="gloo")
    return g


def choose(requested=None):
    """Transport of the device drivers: EIGAMD_SUMMA_TRANSPORT = ipc | rccl | auto (default).  auto: ipc when
    every rank can export and open device memory, else rccl.  Collective."""
    want = (requested or os.environ.get("EIGAMD_SUMMA_TRANSPORT", "auto")).lower()
    if want not in ("ipc", "rccl", "auto"):
        raise ValueError(f"unknown transport {want!r}")
    if want != "auto":
        return want
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return "ipc"   # (nothing travels; no compute unit needs to be set aside)
    ok = 1
    peers = PeerMemory()
    try:
        probe = torch.empty(256, dtype=torch.uint8, device="cuda")
        ranks = list(range(dist.get_world_size()))
        w = Window(probe, ranks, host_group(), peers)
        # read one byte range of the next rank's probe
        nxt = ranks[(dist.get_rank() + 1) % len(ranks)]
        dst = torch.empty(256, dtype=torch.uint8, device="cuda")
        copy_async(dst.data_ptr(), w.ptr[nxt], 256, torch.cuda.current_stream())
        torch.cuda.synchronize()
    except Exception:   # noqa: BLE001  (any failure means: not this transport)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=host_group())
    dist.barrier(group=host_group())   # nobody frees its probe while a neighbour may still read it
    try:
        peers.close()
    except Exception:   # noqa: BLE001
        pass
    return "ipc" if int(flag.item()) == 1 else "rccl"


class MaskedStream:
    """A stream of the library whose kernels leave `leave_free` compute units alone, as a torch stream."""

    def __init__(self, leave_free):
        p = ctypes.c_void_p()
        _lib.check(lib.eigamd_stream_create_cu_mask(int(leave_free), 0, ctypes.byref(p)), "eigamd_stream_create_cu_mask")
        self.raw = p.value
        self.stream = torch.cuda.ExternalStream(self.raw)

    def close(self):
        if self.raw:
            lib.eigamd_stream_destroy(ctypes.c_void_p(self.raw))
            self.raw = None


class PanelStats:
    """Per-panel timing of a distributed product or factorisation, from events on the streams the work runs
    on: time on the copy stream (`comm`), time in the local products (`compute`) and how long the compute
    stream stood waiting for a panel (`exposed`)."""

    def __init__(self):
        self.rows = []   # (comm_begin, comm_end, wait_begin, wait_end, gemm_end)

    def events(self):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        self.rows.append(ev)
        return ev

    def summary(self):
        comm = [a.elapsed_time(b) for a, b, _, _, _ in self.rows]
        exposed = [c.elapsed_time(d) for _, _, c, d, _ in self.rows]
        compute = [d.elapsed_time(e) for _, _, _, d, e in self.rows]
        r = lambda xs: [round(x, 3) for x in xs]
        return {"panels": len(self.rows), "comm_ms": round(sum(comm), 3), "compute_ms": round(sum(compute), 3),
                "exposed_wait_ms": round(sum(exposed), 3), "comm_ms_per_panel": r(comm),
                "exposed_wait_ms_per_panel": r(exposed), "compute_ms_per_panel": r(compute)}
