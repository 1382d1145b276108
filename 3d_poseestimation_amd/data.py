"""Batch feeder for the lifter step: replaces the reference's DataLoader hop
(/root/reference/phase1_lifting/train_1.py:26-31 `DataLoader(training_set, shuffle=True, batch_size=...)`,
:75-81 `.float()`, `.to(device)`) for pose tables that are already arrays.

The MI355X answer to "load a batch" is not to load it: Human3.6M's ~1.5 M training poses are
(17*2 + 17*3) * 4 B = 340 B each, 0.5 GB -- 0.2 % of the 288 GB of HBM.  `PoseFeeder` keeps both
tables resident and a batch is a device-side row gather by the epoch's permutation
(pl_gather_rows2): no per-step H2D copy, no host synchronisation, nothing for the train step to
wait on.  Tables that should not be resident (`resident=False`) stay in pinned host memory and
batches are gathered on the host into pinned staging buffers and copied on a side stream two
batches ahead; the consumer only waits on an event.

Data parallel: every rank draws the same permutation (same seed) and takes its `dp.shard_rows`
slice of each global batch of `batch_size * world` rows.
"""
import torch

from . import _lib
from .dp import shard_rows


def epoch_indices(n_rows, batch_size, *, epoch=0, seed=0, shuffle=True, drop_last=False, rank=0, world=1):
    """The index tensors (int64, CPU) of this rank's batches for one epoch.  A global batch is
    `batch_size * world` consecutive entries of the epoch's permutation; the last one may be short
    (DataLoader's drop_last=False).  With world > 1 every rank must run the same number of steps on the
    same number of rows (the gradient all-reduce and SyncBN are collectives; training-mode BatchNorm needs
    >= 2 rows): the short tail batch is trimmed to a multiple of `world` rows and dropped on EVERY rank when
    that leaves fewer than 2 rows per rank."""
    if n_rows <= 0 or batch_size <= 0 or not (0 <= rank < world):
        raise ValueError("bad feeder geometry")
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(seed) * 1000003 + int(epoch))
        perm = torch.randperm(n_rows, generator=g)
    else:
        perm = torch.arange(n_rows)
    gb = batch_size * world
    out = []
    for start in range(0, n_rows, gb):
        chunk = perm[start:start + gb]
        if chunk.numel() < gb and drop_last:
            break
        if world > 1 and chunk.numel() < gb:
            per = chunk.numel() // world
            if per < 2:
                break
            chunk = chunk[:per * world]
        lo, hi = shard_rows(chunk.numel(), rank, world)
        out.append(chunk[lo:hi])
    return out


def epoch_steps(n_rows, batch_size, *, drop_last=False, world=1):
    """Number of batches epoch_indices() yields -- the same on every rank."""
    gb = batch_size * world
    full, tail = divmod(n_rows, gb)
    if drop_last or tail == 0:
        return full
    return full + (1 if (world == 1 or tail // world >= 2) else 0)


class PoseFeeder:
    """Iterable of (y1 [b,17,2], y2 [b,17,3]) fp32 device batches over one epoch; call
    set_epoch(e) between epochs (as with DistributedSampler) for a fresh permutation."""

    def __init__(self, x2d, y3d, batch_size, *, device="cuda", shuffle=True, seed=0, drop_last=False,
                 rank=0, world=1, resident=True, prefetch=2):
        x2d, y3d = torch.as_tensor(x2d), torch.as_tensor(y3d)
        if x2d.shape[0] != y3d.shape[0] or x2d.shape[0] == 0:
            raise ValueError("x2d and y3d need the same, non-zero number of poses")
        self.n = x2d.shape[0]
        self.shape_a, self.shape_b = tuple(x2d.shape[1:]), tuple(y3d.shape[1:])
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PoseliftError("PoseFeeder feeds an MI355X (ROCm) device; there is no CPU path")
        a = x2d.reshape(self.n, -1).to(torch.float32).contiguous()        # train_1.py:80 .float()
        b = y3d.reshape(self.n, -1).to(torch.float32).contiguous()
        self.wa, self.wb = a.shape[1], b.shape[1]
        self.batch_size, self.shuffle, self.seed, self.drop_last = batch_size, shuffle, seed, drop_last
        self.rank, self.world, self.resident, self.prefetch = rank, world, resident, max(1, int(prefetch))
        self.epoch = 0
        if resident:
            self.a, self.b = a.to(self.device), b.to(self.device)
        else:
            self.a, self.b = a.pin_memory(), b.pin_memory()
            self._copy_stream = torch.cuda.Stream(self.device)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _indices(self):
        return epoch_indices(self.n, self.batch_size, epoch=self.epoch, seed=self.seed, shuffle=self.shuffle,
                             drop_last=self.drop_last, rank=self.rank, world=self.world)

    def __len__(self):
        return epoch_steps(self.n, self.batch_size, drop_last=self.drop_last, world=self.world)

    def __iter__(self):
        return self._iter_resident() if self.resident else self._iter_streamed()

    def _iter_resident(self):
        batches = self._indices()
        if not batches:
            return
        # one H2D of the whole epoch's permutation (8 B per pose), then only device work
        sizes = [t.numel() for t in batches]
        idx_all = torch.cat(batches).to(self.device, non_blocking=True)
        L, at = _lib.lib(), 0
        for nb in sizes:
            oa = torch.empty((nb,) + self.shape_a, dtype=torch.float32, device=self.device)
            ob = torch.empty((nb,) + self.shape_b, dtype=torch.float32, device=self.device)
            with _lib.on_device(self.device):
                _lib.check(L.pl_gather_rows2(self.a.data_ptr(), self.wa, self.b.data_ptr(), self.wb,
                                             idx_all[at:at + nb].data_ptr(), nb, self.n, oa.data_ptr(), ob.data_ptr(),
                                             _lib.current_stream_ptr()), "pl_gather_rows2")
            at += nb
            yield oa, ob

    def _iter_streamed(self):
        batches = self._indices()
        slots, inflight = self.prefetch + 1, []
        stage = [(torch.empty(self.batch_size, self.wa).pin_memory(), torch.empty(self.batch_size, self.wb).pin_memory())
                 for _ in range(slots)]
        free_events = [None] * slots                      # consumer done with the slot's previous batch

        def launch(k, idx):
            slot = k % slots
            nb = idx.numel()
            ha, hb = stage[slot][0][:nb], stage[slot][1][:nb]
            if free_events[slot] is not None:
                free_events[slot].synchronize()           # staging buffer no longer being copied from
            torch.index_select(self.a, 0, idx, out=ha)
            torch.index_select(self.b, 0, idx, out=hb)
            with torch.cuda.stream(self._copy_stream):
                da = ha.to(self.device, non_blocking=True).view((nb,) + self.shape_a)
                db = hb.to(self.device, non_blocking=True).view((nb,) + self.shape_b)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            free_events[slot] = ev
            return da, db, ev

        nxt = 0
        while nxt < len(batches) and len(inflight) < self.prefetch:
            inflight.append(launch(nxt, batches[nxt]))
            nxt += 1
        while inflight:
            da, db, ev = inflight.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev)
            da.record_stream(torch.cuda.current_stream(self.device))
            db.record_stream(torch.cuda.current_stream(self.device))
            if nxt < len(batches):
                inflight.append(launch(nxt, batches[nxt]))
                nxt += 1
            yield da, db
