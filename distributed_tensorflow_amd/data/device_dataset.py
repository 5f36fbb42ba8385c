"""Training split resident in device memory, with on-device batch selection.

The reference uploads every batch from host memory twice per iteration
(``feed_dict`` at /root/reference/distributed.py:138 and :148).  A 55000 x 784
fp32 training set is 172 MB — 0.06 % of one MI355X's HBM — so the whole split
lives on the GPU and the kernels gather the rows of the current batch through
an index vector that the previous step's last kernel wrote.  The host only
uploads one permutation (220 KB) per epoch, ahead of time and in stream order.
"""
from __future__ import annotations

import torch

from .mnist import DataSet, epoch_permutation


_QUARANTINE = []      # device allocations that failed the read-back check; never freed, never used


class DeviceDataset:
    def __init__(self, split: DataSet, device, shuffle_seed: int = 0, shuffle: bool = True):
        self.device = torch.device(device)
        self.images = self._upload(split.images)
        self.labels = self._upload(split.labels)
        self.n = split.num_examples
        self.seed = shuffle_seed
        self.shuffle = shuffle
        # perm[e & 1] holds the row order of epoch e
        self.perm = torch.zeros((2, self.n), dtype=torch.int32, device=self.device) if shuffle else None
        self._resident = [None, None]
        # permutations travel through page-locked staging buffers owned by this object (one per device buffer): the
        # source of the upload then stays valid and in place for as long as the copy engine may be reading it
        self._stage = None
        self._stage_done = [None, None]       # events: the upload that last read each staging buffer has finished
        if shuffle and self.device.type == "cuda":
            self._stage = torch.zeros((2, self.n), dtype=torch.int32).pin_memory()
        # row orders of the OTHER workers' shuffles (enable_peer_shuffles): the N-worker pipelined step gathers every
        # worker's batches on every rank
        self.peer_seeds = None
        self.peer_perm = None
        self._peer_resident = None
        self._peer_stage = None
        self._peer_stage_done = None

    def enable_peer_shuffles(self, seeds, own: int) -> None:
        """Keep the row order of every worker's shuffle resident (``seeds[r]`` = shuffle seed of worker r,
        ``seeds[own]`` this worker's): every worker draws its own batches from its own full copy of the data set
        (/root/reference/distributed.py:38, :137), and with the seeds known every rank can tell which rows they are."""
        seeds = [int(s) for s in seeds]
        if seeds[own] != self.seed:
            raise ValueError("enable_peer_shuffles: seeds[own] is not this data set's shuffle seed")
        self.peer_seeds = seeds
        if not self.shuffle:
            self.peer_perm = [None] * len(seeds)
            return
        self.peer_perm = [self.perm if r == own else torch.zeros((2, self.n), dtype=torch.int32, device=self.device)
                          for r in range(len(seeds))]
        self._peer_resident = [[None, None] for _ in seeds]
        self._peer_own = own
        if self.device.type == "cuda":
            self._peer_stage = torch.zeros((len(seeds), 2, self.n), dtype=torch.int32).pin_memory()
            self._peer_stage_done = [[None, None] for _ in seeds]

    def _upload(self, array) -> torch.Tensor:
        """Host array -> device tensor through a page-locked copy of our own, read by a KERNEL over the host link
        (``ops.upload_from_pinned``), an explicit wait and a read-back check (``ops.h2d_verified``)."""
        t = torch.from_numpy(array)
        if self.device.type != "cuda":
            return t.to(self.device)
        from .. import ops
        # Memory that does not hold what was written to it (seen with recycled exported regions, see
        # ops.h2d_verified) is kept allocated, so that nothing else is placed there, and the split goes to another
        # allocation.
        for _ in range(4):
            out = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            try:
                return ops.h2d_verified(out, t, "dataset split", attempts=2)
            except RuntimeError:
                _QUARANTINE.append(out)
        raise RuntimeError("dataset upload: four different allocations did not hold the uploaded data")

    def _kernel_upload(self, out: torch.Tensor, staged: torch.Tensor, wait: bool = True):
        """Returns ``out`` after waiting for the copy, or (``wait=False``) an event that marks its end: the stream is
        not drained for an epoch's permutation in the middle of a run."""
        from .. import ops
        with torch.cuda.device(self.device):
            if out.element_size() == 4 and ops._lib.kernels_available():
                ops.upload_from_pinned(out, staged)
            else:
                out.copy_(staged, non_blocking=True)
            if not wait:
                ev = torch.cuda.Event()
                ev.record()
                return ev
            torch.cuda.current_stream(self.device).synchronize()
        return out

    def max_steps_per_launch(self, batch: int) -> int:
        """A launch may touch at most two epochs (one per permutation buffer)."""
        return max(1, self.n // batch - 1)

    def prepare(self, first_step: int, n_steps: int, batch: int) -> None:
        """Make the permutations needed by steps ``[first_step, first_step+n_steps]`` resident.

        The last step of a launch already selects the rows of the step after it,
        hence the closed upper bound.  Copies are issued on the current stream,
        i.e. after every earlier launch that might still read the buffer being
        replaced.
        """
        if not self.shuffle:
            return
        e_lo = (first_step * batch) // self.n
        e_hi = ((first_step + n_steps + 1) * batch - 1) // self.n
        if e_hi - e_lo > 1:
            raise ValueError(f"{n_steps} steps of batch {batch} span more than two epochs of {self.n} rows")
        for e in range(e_lo, e_hi + 1):
            if self._resident[e & 1] != e:
                p = torch.from_numpy(epoch_permutation(self.seed, e, self.n))
                if self._stage is not None:
                    # (the launches that may still read the device buffer precede the copy in stream order; the
                    # staging buffer of this slot was last used two epochs ago: wait for that upload's event)
                    ev = self._stage_done[e & 1]
                    if ev is not None:
                        ev.synchronize()
                    # (a plain memory copy through NumPy: a tensor copy wakes the CPU thread pool, which takes tens of
                    # milliseconds when several workers share the host's cores)
                    self._stage[e & 1].numpy()[:] = p.numpy()
                    self._stage_done[e & 1] = self._kernel_upload(self.perm[e & 1], self._stage[e & 1], wait=False)
                else:
                    self.perm[e & 1].copy_(p)
                self._resident[e & 1] = e
            if self.peer_perm is not None:
                for r, seed in enumerate(self.peer_seeds):
                    if r == self._peer_own or self._peer_resident[r][e & 1] == e:
                        continue
                    p = torch.from_numpy(epoch_permutation(seed, e, self.n))
                    if self._peer_stage is not None:
                        ev = self._peer_stage_done[r][e & 1]
                        if ev is not None:
                            ev.synchronize()
                        self._peer_stage[r, e & 1].numpy()[:] = p.numpy()
                        self._peer_stage_done[r][e & 1] = self._kernel_upload(self.peer_perm[r][e & 1],
                                                                              self._peer_stage[r, e & 1], wait=False)
                    else:
                        self.peer_perm[r][e & 1].copy_(p)
                    self._peer_resident[r][e & 1] = e
