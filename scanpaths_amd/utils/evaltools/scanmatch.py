"""ScanMatch scorer with the reference's interface (utils/evaltools/scanmatch.py:38-203), computed by csrc/scanmatch.hip.

    sm = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)
    seq = sm.fixationToSequence(fix).astype(np.int32)           # reference call pattern (utils/evaluation.py:57-59)
    score, align, F = sm.match(seq_a, seq_b)

plus the batched form the validation / RL-reward loops need (every sampled scanpath against every human scanpath):

    seqs, lens = sm.sequences(list_of_fixation_arrays)           # device int32 [n, ld], [n]
    scores = sm.match_pairs(seqs, lens, seqs2, lens2, pairs)     # float64 [npairs], one wavefront per pair

Arithmetic is float64 in the reference's operation order: scores, sequences, F and the alignment are bit-exact with
the numpy implementation (tests/test_scanmatch_gpu.py against tests/golden/scanmatch.npz).  There is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import hip
from ...hip import check, ptr


MAX_SYMBOLS = 1 << 18      # hard cap per sequence (3.6 h of fixation at TempBin = 50 ms): bounds the symbol / scratch buffers


class SequenceTooLong(ValueError):
    """a scanpath maps to more than MAX_SYMBOLS symbols (or to a non-finite count)"""


class ScanMatch(object):
    def __init__(self, **kw):
        self.Xres = 1024
        self.Yres = 768
        self.Xbin = 8
        self.Ybin = 6
        self.Threshold = 3.5
        self.GapValue = 0.0
        self.TempBin = 0.0
        self.Offset = (0, 0)
        for k, v in kw.items():
            if k not in ("Xres", "Yres", "Xbin", "Ybin", "Threshold", "GapValue", "TempBin", "Offset"):
                raise ValueError('Unknown parameter: %s.' % k)
            setattr(self, k, v)
        if not torch.cuda.is_available():
            raise hip.HipError("scanpaths_amd ScanMatch runs on a HIP device only (no CPU path)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._mask_dev: Optional[torch.Tensor] = None      # custom mask (maskFromArray); None -> arithmetic grid
        self.CreateSubMatrix()
        self.GridMask()

    # ---- tables ------------------------------------------------------------------------------------------
    def CreateSubMatrix(self, Threshold=None):
        if Threshold is not None:
            self.Threshold = Threshold
        nb = self.Xbin * self.Ybin
        self._sub = torch.empty((nb, nb), dtype=torch.float64, device=self.device)
        self._maxsub = torch.empty(1, dtype=torch.float64, device=self.device)
        check(hip.lib().sp_scanmatch_submatrix(int(self.Xbin), int(self.Ybin), float(self.Threshold), ptr(self._sub),
                                               ptr(self._maxsub), hip.stream()), "sp_scanmatch_submatrix")

    @property
    def SubMatrix(self) -> np.ndarray:
        return self._sub.cpu().numpy()

    def GridMask(self):
        """the reference materialises a [Yres, Xres] lookup table; the kernels bin arithmetically (same values)"""
        self._mask_dev = None

    @property
    def mask(self) -> np.ndarray:
        if self._mask_dev is not None:
            return self._mask_dev.cpu().numpy().astype(np.float64)
        ys, xs = torch.meshgrid(torch.arange(self.Yres, dtype=torch.float64), torch.arange(self.Xres, dtype=torch.float64),
                                indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).contiguous()
        saved = self.Offset, self.TempBin
        self.Offset, self.TempBin = (0, 0), 0.0
        try:
            seq, _ = self._sequences_dev(pix.to(self.device), torch.arange(pix.shape[0], device=self.device),
                                         torch.ones(pix.shape[0], dtype=torch.int32, device=self.device), ld=1)
        finally:
            self.Offset, self.TempBin = saved
        return seq.view(self.Yres, self.Xres).cpu().numpy().astype(np.float64)

    def maskFromArray(self, array):
        a = np.ascontiguousarray(np.asarray(array))
        if a.shape != (self.Yres, self.Xres):
            raise ValueError(f"mask must be [Yres, Xres] = {(self.Yres, self.Xres)}, got {a.shape}")
        self._mask_dev = torch.from_numpy(a.astype(np.int32)).to(self.device)

    def subMatrixFromArray(self, array):
        # the reference assigns to a misspelt attribute (scanmatch.py:202-203), i.e. this call has no effect there either
        self.SubMarix = array

    # ---- fixations -> symbol strings ---------------------------------------------------------------------------
    def _sequences_dev(self, fix: torch.Tensor, start: torch.Tensor, count: torch.Tensor, ld: Optional[int] = None
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
        L = hip.lib()
        fix = fix.to(self.device, torch.float64).contiguous()
        start = start.to(self.device, torch.int64).contiguous()
        count = count.to(self.device, torch.int32).contiguous()
        nsp, ncol = int(count.numel()), int(fix.shape[1])
        lens = torch.empty(nsp, dtype=torch.int32, device=self.device)
        args = (int(self.Xres), int(self.Yres), int(self.Xbin), int(self.Ybin), float(self.Offset[0]), float(self.Offset[1]),
                float(self.TempBin), ptr(self._mask_dev))
        if ld is None:                              # sizing pass (one host sync, as the reference's python lists)
            check(L.sp_scanmatch_sequences(ptr(fix), ncol, ptr(start), ptr(count), nsp, *args, 0, None, ptr(lens), hip.stream()),
                  "sp_scanmatch_sequences")
            ld = max(1, int(lens.max().item()))
        if ld > MAX_SYMBOLS or ld < 0:      # longer than the LDS kernel's sp_scanmatch_max_len() is fine: match_pairs switches kernels
            raise SequenceTooLong(f"sequence of {ld} symbols exceeds the cap of {MAX_SYMBOLS}")
        seq = torch.zeros((nsp, ld), dtype=torch.int32, device=self.device)
        check(L.sp_scanmatch_sequences(ptr(fix), ncol, ptr(start), ptr(count), nsp, *args, ld, ptr(seq), ptr(lens), hip.stream()),
              "sp_scanmatch_sequences")
        return seq, lens

    def sequences(self, scanpaths: Sequence[np.ndarray]) -> Tuple[torch.Tensor, torch.Tensor]:
        """list of [n_k, 2|3] fixation arrays -> (symbols int32 [n, ld] on the device, lengths int32 [n])"""
        arrs = [np.asarray(a, dtype=np.float64).reshape(len(a), -1) for a in scanpaths]
        ncol = arrs[0].shape[1] if arrs else 3
        if any(a.shape[1] != ncol for a in arrs):
            raise ValueError("all scanpaths need the same number of columns")
        if self.TempBin != 0 and ncol < 3:
            raise IndexError("TempBin != 0 needs a duration column")          # the reference raises IndexError on d[:, 2]
        count = torch.tensor([a.shape[0] for a in arrs], dtype=torch.int32)
        start = torch.cumsum(count.to(torch.int64), 0) - count.to(torch.int64)
        cat = np.concatenate(arrs, 0) if arrs else np.zeros((0, ncol))
        if cat.shape[0] == 0:
            cat = np.zeros((1, ncol))
        return self._sequences_dev(torch.from_numpy(cat), start, count)

    def fixationToSequence(self, data):
        data = np.asarray(data, dtype=np.float64)
        seq, lens = self.sequences([data])
        return seq[0, :int(lens[0].item())].cpu().numpy().astype(np.float64)      # the reference returns a float array

    # ---- Needleman-Wunsch --------------------------------------------------------------------------------------
    def match_pairs(self, seqA: torch.Tensor, lenA: torch.Tensor, seqB: torch.Tensor, lenB: torch.Tensor,
                    pairs: Optional[torch.Tensor] = None) -> torch.Tensor:
        """scores float64 [npairs] on the device; pairs int32 [npairs, 2] = (row of seqA, row of seqB); None: row k with row k"""
        seqA, seqB = seqA.contiguous(), seqB.contiguous()
        if pairs is None:
            if seqA.shape[0] != seqB.shape[0]:
                raise ValueError("pairs=None needs equally many sequences on both sides")
            npairs = seqA.shape[0]
        else:
            pairs = pairs.to(self.device, torch.int32).contiguous()
            npairs = pairs.shape[0]
        scores = torch.empty(npairs, dtype=torch.float64, device=self.device)
        if npairs == 0:
            return scores
        L = hip.lib()
        if max(seqA.shape[1], seqB.shape[1]) <= L.sp_scanmatch_max_len():
            check(L.sp_scanmatch_score(ptr(seqA), ptr(lenA), seqA.shape[1], ptr(seqB), ptr(lenB), seqB.shape[1], ptr(pairs),
                                       npairs, ptr(self._sub), self._sub.shape[0], ptr(self._maxsub), float(self.GapValue),
                                       ptr(scores), hip.stream()), "sp_scanmatch_score")
        else:       # rare: a heavy-tailed sampled duration; the strip column moves from LDS to a global scratch row per pair
            ws = torch.empty(L.sp_scanmatch_score_long_workspace(seqA.shape[1], npairs), dtype=torch.uint8, device=self.device)
            check(L.sp_scanmatch_score_long(ptr(seqA), ptr(lenA), seqA.shape[1], ptr(seqB), ptr(lenB), seqB.shape[1], ptr(pairs),
                                            npairs, ptr(self._sub), self._sub.shape[0], ptr(self._maxsub), float(self.GapValue),
                                            ptr(scores), ptr(ws), hip.stream()), "sp_scanmatch_score_long")
        return scores

    def match_all(self, scanpaths_a: Sequence[np.ndarray], scanpaths_b: Sequence[np.ndarray]) -> np.ndarray:
        """score matrix [len(a), len(b)] of every scanpath in a against every scanpath in b"""
        sa, la = self.sequences(scanpaths_a)
        sb, lb = self.sequences(scanpaths_b)
        ia, ib = torch.meshgrid(torch.arange(sa.shape[0]), torch.arange(sb.shape[0]), indexing="ij")
        pairs = torch.stack([ia.reshape(-1), ib.reshape(-1)], 1).to(torch.int32)
        return self.match_pairs(sa, la, sb, lb, pairs).view(sa.shape[0], sb.shape[0]).cpu().numpy()

    def match(self, A, B):
        """(score, alignment [steps, 2] with -1 for gaps, F transposed) as scanmatch.py:137-197"""
        A = np.asarray(A).astype(np.int32).reshape(-1)
        B = np.asarray(B).astype(np.int32).reshape(-1)
        n, m = len(A), len(B)
        nb = self._sub.shape[0]
        if (n and (A.min() < 0 or A.max() >= nb)) or (m and (B.min() < 0 or B.max() >= nb)):
            raise IndexError("symbol outside the substitution matrix")
        dev = self.device
        Ad = torch.from_numpy(A).to(dev) if n else None
        Bd = torch.from_numpy(B).to(dev) if m else None
        Fw = torch.empty((n + 1) * (m + 1), dtype=torch.float64, device=dev)
        Ft = torch.empty((m + 1, n + 1), dtype=torch.float64, device=dev)
        al = torch.empty((max(1, n + m), 2), dtype=torch.float64, device=dev)
        nal = torch.empty(1, dtype=torch.int32, device=dev)
        sc = torch.empty(1, dtype=torch.float64, device=dev)
        check(hip.lib().sp_scanmatch_align(ptr(Ad), n, ptr(Bd), m, ptr(self._sub), nb, ptr(self._maxsub), float(self.GapValue),
                                           ptr(Fw), ptr(Ft), ptr(al), ptr(nal), ptr(sc), hip.stream()), "sp_scanmatch_align")
        steps = int(nal.item())
        return float(sc.item()), al[:steps].cpu().numpy(), Ft.cpu().numpy()
