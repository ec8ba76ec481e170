"""Several GPUs, one process per GPU: thin Python side of the distributed-archive calls of include/zra_hip.h
(zra_amd/csrc/zra_comm.hip). Frames are independent, so rank r of W owns the contiguous frame range [F*r/W, F*(r+1)/W); the only
exchanges are the all-gather of the per-frame sizes after compression, and — for serving — query slices to the owners and the decoded
bytes back. All of that lives behind the C ABI; this module only

  * creates the communicator: RCCL called directly by the library (`Comm.rccl`: the 128-byte id travels over torch.distributed's
    store), or the host transport, whose two callbacks are implemented here with torch.distributed (`Comm.torch_dist`: gloo in the
    CPU tests and the one-GPU dry runs, nccl = RCCL on GPU tensors otherwise);
  * exposes the router (`route_queries`, pure host arithmetic — no GPU needed).
"""
import ctypes
import sys

import numpy as np

import zra_amd as Z


def shard_range(nframes, rank, world):
    """Frames [lo, hi) owned by `rank` (ZraHipShardRange)."""
    lo, hi = ctypes.c_uint64(0), ctypes.c_uint64(0)
    Z.load().ZraHipShardRange(nframes, rank, world, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def owner_of_frame(nframes, world, frame):
    return Z.load().ZraHipOwnerOfFrame(nframes, world, frame)


def _u64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def route_queries(uncompressed_size, frame_size, world, offsets, sizes):
    """Cuts queries at ownership boundaries (ZraHipRouteQueries). Returns (slices, per_owner_count); slices is a structured array
    with fields owner, query, offset, size, within, grouped by owner."""
    L = Z.load()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64); sizes = np.ascontiguousarray(sizes, dtype=np.uint64)
    cap = len(offsets) + world + 16
    per = np.zeros(world, dtype=np.uint64)
    for _ in range(2):
        buf = (Z.ZraHipSlice * cap)()
        n = ctypes.c_size_t(0)
        st = L.ZraHipRouteQueries(uncompressed_size, frame_size, world, _u64p(offsets), _u64p(sizes), len(offsets), buf, cap, ctypes.byref(n), _u64p(per))
        if st.zra == 6:
            cap = n.value
            continue
        Z._chk(st, "ZraHipRouteQueries")
        break
    out = np.zeros(n.value, dtype=[("owner", np.uint32), ("query", np.uint64), ("offset", np.uint64), ("size", np.uint64), ("within", np.uint64)])
    for i in range(n.value):
        out[i] = (buf[i].owner, buf[i].query, buf[i].offset, buf[i].size, buf[i].within)
    return out, per


class Shard:
    """This rank's part of a distributed archive (ZraHipShard): complete header + seek table, own frames' compressed bytes."""

    def __init__(self, handle):
        self.h = handle
        self.L = Z.load()

    def header(self):
        n = self.L.ZraHipShardHeaderSize(self.h)
        b = ctypes.create_string_buffer(n)
        self.L.ZraHipShardGetHeader(self.h, b)
        return b.raw[:n]

    def archive_size(self):
        return self.L.ZraHipShardArchiveSize(self.h)

    def body(self):
        """(device pointer, offset inside the archive's body, bytes) of this rank's frames"""
        p, base, n = ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.L.ZraHipShardGetBody(self.h, ctypes.byref(p), ctypes.byref(base), ctypes.byref(n))
        return p.value or 0, base.value, n.value

    def close(self):
        if self.h:
            self.L.ZraHipShardDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():          # at interpreter exit the HIP runtime may already be gone
                self.close()
        except Exception:
            pass


class Comm:
    """ZraHipComm. All methods are collective."""

    def __init__(self, handle, rank, world, keep=None, engine=None):
        self.h, self.rank, self.world, self._keep = handle, rank, world, keep
        self.engine = engine
        self.L = Z.load()

    def _order(self):
        # device buffers handed to the collective calls are usually torch tensors produced asynchronously on torch's stream: the
        # engine's streams wait for it first (event wait, no host sync) — the same plumbing as Engine._order (found by the 8-rank dry
        # run: content checksums taken from an input that torch was still writing)
        if self.engine is not None:
            self.engine._order()

    # ---- construction
    @classmethod
    def rccl(cls, engine, rank, world, group=None):
        """RCCL called by the library itself; the unique id is broadcast over torch.distributed (any backend)."""
        import torch.distributed as dist
        L = Z.load()
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            Z._chk(L.ZraHipCommGetUniqueId(ident), "ZraHipCommGetUniqueId")
        box = [ident.raw if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
        h = ctypes.c_void_p()
        Z._chk(L.ZraHipCommCreateRccl(ctypes.byref(h), engine.h, ident, rank, world), "ZraHipCommCreateRccl")
        return cls(h, rank, world, engine=engine)

    @classmethod
    def torch_dist(cls, engine, group=None):
        """Host transport over torch.distributed: gloo moves host buffers as they are, nccl (= RCCL) gets them as device tensors."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"

        def as_tensor(ptr, n):
            t = torch.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=torch.uint8) if n else torch.empty(0, dtype=torch.uint8)
            return t

        def allgather(user, send, recv, nbytes):
            try:
                s = as_tensor(send, nbytes); r = as_tensor(recv, nbytes * world)
                if on_gpu:
                    sd = s.cuda(); outs = [torch.empty_like(sd) for _ in range(world)]
                    dist.all_gather(outs, sd, group=group)
                    r.copy_(torch.cat(outs).cpu())
                else:
                    outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(outs, s.clone(), group=group)
                    r.copy_(torch.cat(outs))
                return 0
            except Exception as e:          # a Python exception must not unwind through the C frames
                print("zra_amd.sharding allgather failed:", repr(e))
                return 1

        def exchange(user, ns, sp, sb, sn, nr, rp, rb, rn):
            try:
                ops, back = [], []
                for i in range(nr):
                    host = as_tensor(rb[i], rn[i])
                    t = torch.empty(rn[i], dtype=torch.uint8, device="cuda") if on_gpu else host
                    back.append((host, t))
                    ops.append(dist.P2POp(dist.irecv, t, rp[i], group))
                for i in range(ns):
                    t = as_tensor(sb[i], sn[i])
                    ops.append(dist.P2POp(dist.isend, t.cuda() if on_gpu else t.clone(), sp[i], group))
                if ops:
                    for q in dist.batch_isend_irecv(ops):
                        q.wait()
                if on_gpu:
                    for host, t in back:
                        host.copy_(t.cpu())
                return 0
            except Exception as e:
                print("zra_amd.sharding exchange failed:", repr(e))
                return 1

        tr = Z.ZraHipHostTransport(None, Z.ALLGATHER_FN(allgather), Z.EXCHANGE_FN(exchange))
        h = ctypes.c_void_p()
        Z._chk(Z.load().ZraHipCommCreateHost(ctypes.byref(h), engine.h, ctypes.byref(tr), rank, world), "ZraHipCommCreateHost")
        return cls(h, rank, world, keep=tr, engine=engine)

    # ---- collectives
    def compress(self, d_local, local_bytes, total_bytes, level, frame_size, checksum=True):
        """ZraHipCommCompress: d_local = device pointer of this rank's frames' bytes. Returns a Shard."""
        sh = ctypes.c_void_p()
        self._order()
        Z._chk(self.L.ZraHipCommCompress(self.h, d_local, local_bytes, total_bytes, level, frame_size, checksum, ctypes.byref(sh)), "ZraHipCommCompress")
        return Shard(sh)

    def gather_archive(self, shard, root=0, d_archive=0, capacity=0):
        """ZraHipCommGatherArchive: the archive in one piece on `root` (device pointer + capacity there). Returns its size on the root."""
        n = ctypes.c_size_t(0)
        self._order()
        Z._chk(self.L.ZraHipCommGatherArchive(self.h, shard.h, root, d_archive, capacity, ctypes.byref(n)), "ZraHipCommGatherArchive")
        return n.value

    def use_own_stream(self):
        """ZraHipCommUseOwnStream: this (second) communicator's exchanges run on a stream of its own, beside the engine's."""
        Z._chk(self.L.ZraHipCommUseOwnStream(self.h), "ZraHipCommUseOwnStream")
        return self

    def gather_archive_begin(self, shard, root=0, d_archive=0, capacity=0):
        """ZraHipCommGatherArchiveBegin: the gather starts on this communicator's own stream and the call returns; serve on ANOTHER
        communicator meanwhile, then gather_archive_end()."""
        self._order()
        Z._chk(self.L.ZraHipCommGatherArchiveBegin(self.h, shard.h, root, d_archive, capacity), "ZraHipCommGatherArchiveBegin")

    def gather_archive_end(self):
        n = ctypes.c_size_t(0)
        Z._chk(self.L.ZraHipCommGatherArchiveEnd(self.h, ctypes.byref(n)), "ZraHipCommGatherArchiveEnd")
        return n.value

    def loopback(self, d_src, d_dst, nbytes):
        """ZraHipCommLoopback: nbytes from d_src to d_dst through the RCCL point-to-point path, this rank to itself (diagnostic)."""
        self._order()
        Z._chk(self.L.ZraHipCommLoopback(self.h, d_src, d_dst, nbytes), "ZraHipCommLoopback")

    def serve(self, shard, offsets, sizes, out_offsets, d_out):
        """ZraHipCommServe: this rank's queries over the whole uncompressed range; answers at d_out + out_offsets[q]."""
        o = np.ascontiguousarray(offsets, dtype=np.uint64); s = np.ascontiguousarray(sizes, dtype=np.uint64); d = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        self._order()
        Z._chk(self.L.ZraHipCommServe(self.h, shard.h, _u64p(o), _u64p(s), _u64p(d), len(o), d_out), "ZraHipCommServe")

    def close(self):
        if self.h:
            self.L.ZraHipCommDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():          # at interpreter exit the HIP runtime may already be gone
                self.close()
        except Exception:
            pass
