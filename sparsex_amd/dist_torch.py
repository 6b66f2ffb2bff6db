"""A transport for ``spx_hip_mat_dist_attach`` made of ``torch.distributed`` calls.

The product's own transport is RCCL point-to-point inside ``libsparsex.so``
(``spx_hip_transport_rccl``).  This one exists so that the same exchange plan can
run where RCCL cannot -- the CPU test-suite and the single-GPU box, where several
processes share a device and talk over gloo -- and as a plainly reported
stand-by for ``bench.py``.  It stages every segment through tensors of its own:
host tensors for gloo, device tensors for nccl.
"""
import ctypes as C

import numpy as np

from .api import CallbackTransport

_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


def torch_transport(rank, world, staging_device="cpu"):
    """CallbackTransport over the default process group (all_to_all_single)."""
    import torch
    import torch.distributed as dist

    def a2a(send_t, sc, rc):
        recv_t = torch.empty(int(sum(rc)), dtype=send_t.dtype, device=send_t.device)
        dist.all_to_all_single(recv_t, send_t, [int(v) for v in rc], [int(v) for v in sc])
        return recv_t

    def counts(cnt):
        c = [int(v) for v in cnt]
        c[rank] = 0
        return c

    def host(send, soff, scnt, recv, roff, rcnt):
        sc, rc = counts(scnt), counts(rcnt)
        parts = [send[int(soff[q]):int(soff[q]) + sc[q]].astype(np.int64) for q in range(world)]
        send_t = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64))
        if str(staging_device) != "cpu":
            send_t = send_t.to(staging_device)
        got = a2a(send_t, sc, rc).cpu().numpy().astype(np.uint64)
        k = 0
        for q in range(world):
            recv[int(roff[q]):int(roff[q]) + rc[q]] = got[k:k + rc[q]]
            k += rc[q]

    def device(send_ptr, soff, scnt, recv_ptr, roff, rcnt, stream):
        hip = _hiprt()
        sc, rc = counts(scnt), counts(rcnt)
        on_host = str(staging_device) == "cpu"
        send_t = torch.empty(sum(sc), dtype=torch.float64, device=staging_device)
        kind_out = 2 if on_host else 3          # hipMemcpyDeviceToHost / DeviceToDevice
        kind_in = 1 if on_host else 3           # hipMemcpyHostToDevice / DeviceToDevice
        k = 0
        for q in range(world):
            if sc[q]:
                rcode = hip.hipMemcpyAsync(send_t.data_ptr() + 8 * k, send_ptr + 8 * int(soff[q]), 8 * sc[q],
                                           kind_out, stream)
                assert rcode == 0, "hipMemcpyAsync failed (%d)" % rcode
            k += sc[q]
        assert hip.hipStreamSynchronize(stream) == 0
        recv_t = a2a(send_t, sc, rc)
        if not on_host:
            torch.cuda.synchronize()
        k = 0
        for q in range(world):
            if rc[q]:
                rcode = hip.hipMemcpyAsync(recv_ptr + 8 * int(roff[q]), recv_t.data_ptr() + 8 * k, 8 * rc[q],
                                           kind_in, stream)
                assert rcode == 0, "hipMemcpyAsync failed (%d)" % rcode
            k += rc[q]
        assert hip.hipStreamSynchronize(stream) == 0      # recv_t must outlive the copies

    return CallbackTransport(rank, world, host, device)
