"""Attach a communicator to an engine from a torch.distributed process group (one process per GPU).

The collective itself lives in the library (include/scema_md.h, "multi-GPU"): `attach_rccl` only carries the 128-byte
RCCL id from rank 0 to the others (what MPI_Bcast does in SCEMa's host program), `attach_gloo` plugs gloo in as the
engine's host transport (CPU-side collective: tests, or several ranks sharing one GPU)."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import capi


def attach_rccl(eng: "capi.Engine", rank: int, world: int) -> None:
    uid = [eng.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    eng.comm_init_rccl(uid[0], rank, world)


def attach_gloo(eng: "capi.Engine", rank: int, world: int) -> None:
    def _ag(b: bytes) -> bytes:
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return torch.cat(parts).numpy().tobytes()

    def _send(b: bytes, dst: int) -> None:
        dist.send(torch.frombuffer(bytearray(b), dtype=torch.uint8), dst)

    def _recv(nbytes: int, src: int) -> bytes:
        t = torch.empty(nbytes, dtype=torch.uint8)
        dist.recv(t, src)
        return t.numpy().tobytes()

    eng.comm_init_host(rank, world, _ag, _send, _recv)
