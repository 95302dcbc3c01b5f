"""Banded view of a (K x N) filterbank for the fused STFT -> mel kernel and the stand-alone banded projection.

Mel filterbanks are banded: column n is zero outside a short run of frequency rows.  The fused
forward kernel exploits exactly that (and nothing else): after the FFT a wave holds |X| of one frame
in its LDS slab, and every lane walks the band of one filter per *pass* (N <= 64 * passes), four
bins per step (`ds_read_b128` of the magnitudes, `ds_read_b128` of the weights).  This module turns
the dense bank into the three tables that walk needs -- built on the host, once per bank version,
cached by the caller:

  lane_filter[q*64 + l]   filter summed by lane l in pass q (-1: none)
  lane_start [q*64 + l]   first bin of that lane's walk (multiple of 4)
  weights                 pass-major, quad-major, lane-minor: the 4 weights lane l multiplies in step j
                          of pass q sit at float offset ((quad_base[q] + j) * 64 + l) * 4

so consecutive lanes read consecutive 16-byte slots of the weight table (conflict-free for any band
length), and the magnitudes are the only access whose banking depends on the bank itself.  A
`ds_read_b128` is served in four groups of 16 lanes; two lanes of a group collide when they read
different 16-byte slots that are congruent mod 256 bytes.  The lane assignment below therefore picks,
per group, filters whose band starts fall on distinct slots (mod 16); a filter shorter than its
pass's walk may start up to `slack` quads early (zero weights in front), which is what makes a
collision-free choice exist in practice.  `lds_read_cycles()` evaluates the result with the same
banking model, so the property is testable without a GPU.

A bank that is not banded enough reports `eligible = False` and the dense MFMA projection
(`at_mel_project`) is used instead.  `eligible` is what the stand-alone projection (mel_banded.hip) takes: rows
of up to 2112 bins (n_fft <= 4096), up to 40 passes, bands of up to 512 bins, as long as the weight table, the lane
tables and four rows fit the 160 KB of LDS.  `fusable` is the tighter set the fused n_fft = 1024 epilogue
(stft1024.hip) takes: 16 passes, bands of 128 bins, a 32 KB weight table next to the FFT slabs.
"""
import numpy as np
import torch

MAX_PASSES = 40       # filters per lane (kMaxBandPasses in band_bank.h: N <= 2560; the default bank has n_fft/2+1)
MAX_BAND = 512        # longest band a lane will walk (kMaxWalk in mel_banded.hip)
MAX_ROW = 2112        # longest input row (kMaxRowK)
LDS_BUDGET = 160 * 1024
FUSED_MAX_PASSES = 16     # the fused epilogue of stft1024.hip: kMaxFusedPasses,
FUSED_MAX_BAND = 128      # its longest walk,
FUSED_TABLE_FLOATS = 8192  # its LDS copy of the weights (kMaxBandFloats)
FUSED_ROW_FLOATS = 640    # and the |X| row a walk may run over
# lanes served together by one LDS cycle of a ds_read_b128 (MI355X_MICROARCH.md, LDS table)
B128_GROUPS = (
    (0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27),
    (4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31),
    (32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59),
    (36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63),
)


def _pass_cycles(lane_quad, walk_quads):
    """LDS cycles of one pass's magnitude reads (ds_read_b128 banking model) and the conflict-free minimum."""
    cycles = 0
    for j in range(walk_quads):
        for lanes in B128_GROUPS:
            per_residue = {}
            for lane in lanes:
                per_residue.setdefault(int(lane_quad[lane] + j) % 16, set()).add(int(lane_quad[lane] + j))
            cycles += max(len(v) for v in per_residue.values())
    return cycles, walk_quads * len(B128_GROUPS)


def _assign_pass_once(filters, first_quad, n_quads, walk_quads, rng):
    lane_filter = np.full(64, -1, np.int64)
    lane_quad = np.zeros(64, np.int64)
    taken = [dict() for _ in B128_GROUPS]       # residue (slot mod 16) -> absolute slot
    used = [0] * len(B128_GROUPS)
    slack = {f: min(walk_quads - n_quads[f], first_quad[f]) for f in filters}
    noise = {f: (rng.random() if rng is not None else 0.0) for f in filters}
    for f in sorted(filters, key=lambda f: (slack[f], noise[f], f)):     # least freedom first
        choice = None
        shifts = list(range(slack[f] + 1))
        if rng is not None:
            rng.shuffle(shifts)
        for d in shifts:
            quad = first_quad[f] - d
            for g in sorted(range(len(B128_GROUPS)), key=lambda g: (used[g], noise[f] * (g + 1) % 1.0)):
                if used[g] < 16 and taken[g].get(quad % 16, quad) == quad:
                    choice = (g, quad)
                    break
            if choice:
                break
        if choice is None:                                      # collision: least loaded group, no shift
            g = min((g for g in range(len(B128_GROUPS)) if used[g] < 16), key=lambda g: used[g])
            choice = (g, first_quad[f])
        g, quad = choice
        lane = B128_GROUPS[g][used[g]]
        used[g] += 1
        taken[g].setdefault(quad % 16, quad)
        lane_filter[lane], lane_quad[lane] = f, quad
    # idle lanes still issue the reads: park each on a slot nobody else in its group uses
    for g, lanes in enumerate(B128_GROUPS):
        for lane in lanes[used[g]:]:
            free = [r for r in range(16) if r not in taken[g]]
            quad = free[0] if free else 0
            taken[g].setdefault(quad % 16, quad)
            lane_quad[lane] = quad
    return lane_filter, lane_quad


def _assign_pass(filters, first_quad, n_quads, walk_quads, tries=64):
    """Place the filters of one pass on the 64 lanes.  Returns (lane_filter[64], lane_quad[64]):
    lane_quad is the 16-byte slot index (bin // 4) at which the lane's walk starts.  Greedy, least
    freedom first; if the plain order leaves collisions, a few seeded random tie-breaks are tried and
    the cheapest placement under the banking model is kept (deterministic for a given bank)."""
    best = None
    rng = None
    for attempt in range(tries):
        lf, lq = _assign_pass_once(filters, first_quad, n_quads, walk_quads, rng)
        cycles, ideal = _pass_cycles(lq, walk_quads)
        if best is None or cycles < best[0]:
            best = (cycles, lf, lq)
        if cycles == ideal:
            break
        rng = np.random.default_rng(1234 + attempt)
    return best[1], best[2]


class BandedBank:
    def __init__(self, bank: torch.Tensor):
        """bank: (K, N) or (1, K, N) dense filterbank (any device)."""
        b = bank.detach().reshape(bank.shape[-2], bank.shape[-1]).float().cpu().numpy()
        K, N = b.shape
        self.K, self.N = K, N
        nz = b != 0
        has = nz.any(0)
        first = np.where(has, nz.argmax(0), 0)
        last = np.where(has, K - 1 - nz[::-1].argmax(0), -1)
        first_quad = (first // 4).astype(np.int64)            # walks start on a 16-byte boundary
        n_quads = np.where(has, (last - first_quad * 4) // 4 + 1, 0).astype(np.int64)
        self.lmax = int((n_quads * 4).max()) if N else 0
        self.n_passes = (N + 63) // 64
        self.executed_macs = int(np.where(has, last - first + 1, 0).sum())   # useful multiply-adds per frame
        self.eligible = self.fusable = self.fusable2048 = self.fusable512 = False
        if not (0 < N and self.n_passes <= MAX_PASSES and self.lmax <= MAX_BAND and K <= MAX_ROW):
            return
        # passes: filters sorted by band length, 64 per pass, so that every pass walks bands of similar length
        order = np.argsort(-n_quads, kind="stable")
        pass_len = np.zeros(MAX_PASSES, np.int32)
        lane_filter = np.full(self.n_passes * 64, -1, np.int32)
        lane_start = np.zeros(self.n_passes * 64, np.int32)
        tables = []
        for q in range(self.n_passes):
            chunk = [int(f) for f in order[q * 64:(q + 1) * 64]]
            walk = int(max(n_quads[f] for f in chunk))
            pass_len[q] = 4 * walk
            lf, lq = _assign_pass(chunk, first_quad, n_quads, walk)
            lane_filter[q * 64:(q + 1) * 64] = lf
            lane_start[q * 64:(q + 1) * 64] = 4 * lq
            w = np.zeros((walk, 64, 4), np.float32)           # [step][lane][4 bins]
            for lane in range(64):
                f = lf[lane]
                if f < 0:
                    continue
                lo = 4 * int(lq[lane])
                hi = min(lo + 4 * walk, K)
                col = np.zeros(4 * walk, np.float32)
                col[:hi - lo] = b[lo:hi, f]
                assert not b[:lo, f].any() and not b[hi:, f].any()
                w[:, lane, :] = col.reshape(walk, 4)
            tables.append(w.reshape(-1))
        weights = np.concatenate(tables) if tables else np.zeros(0, np.float32)
        # the LDS plan of mel_banded.hip (launch_banded): table + lane tables + at least four rows
        segs = (K + 63) // 64
        kernel_segs = segs if segs <= 10 else (17 if segs <= 17 else 33)
        row_floats = -(-max(max(K, 64) + int(pass_len.max()), 64 * kernel_segs) // 64) * 64
        if weights.size * 4 + 2 * 64 * self.n_passes * 4 + 4 * row_floats * 4 > LDS_BUDGET:
            return
        # the features-only n_fft = 512 kernel (stft512.hip): two LDS rows per wave, dynamic LDS within 48 KB
        row512 = -(-(257 + int(pass_len.max())) // 64) * 64
        self.fusable512 = (K == 257 and self.n_passes <= FUSED_MAX_PASSES and self.lmax <= FUSED_MAX_BAND
                           and weights.size <= FUSED_TABLE_FLOATS
                           and 4 * (8 * row512 + weights.size) + 8 * 64 * self.n_passes <= 48 * 1024)
        self.eligible = True
        # (K == 513: the fused n_fft = 1024 epilogue walks a 513-bin row; a bank built for another n_fft would be walked
        #  against the wrong spectrum without any shape check on the way -- ADVICE r2)
        self.fusable = (K == 513 and self.n_passes <= FUSED_MAX_PASSES and self.lmax <= FUSED_MAX_BAND
                        and weights.size <= FUSED_TABLE_FLOATS
                        and int(lane_start.max()) + int(pass_len.max()) <= FUSED_ROW_FLOATS)
        # the features-only n_fft = 2048 kernel (stft2048.hip): same walk limits, its own LDS row (sized per launch)
        # and the whole workgroup within 80 KB of LDS (launch_stft2048_mel)
        row2k = -(-(1025 + int(pass_len.max())) // 64) * 64
        self.fusable2048 = (K == 1025 and self.n_passes <= FUSED_MAX_PASSES and self.lmax <= FUSED_MAX_BAND
                            and weights.size <= FUSED_TABLE_FLOATS
                            and 8 * (4 * 568 + 22 * 64 + 1024) + 4 * (4 * row2k + weights.size) + 8 * 64 * self.n_passes
                            <= 80 * 1024)          # (the window joins them in LDS when another 8 KB fit: launcher)
        self.pass_len = pass_len                              # host array handed to the C ABI
        # the same bank by filter, for the one-pass PolarIF.forward (at_polarif_forward): first bin, bins up to the last
        # non-zero weight, offset of that run in one flat weight array
        length = np.where(has, last - first + 1, 0).astype(np.int32)
        offset = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
        flat = np.concatenate([b[first[f]:first[f] + length[f], f] for f in range(N)] or [np.zeros(0, np.float32)])
        self._by_filter_host = (first.astype(np.int32), length, offset, flat.astype(np.float32))
        self.walked_macs = 64 * int(pass_len.sum())           # multiply-adds issued per frame (incl. zeros)
        self._host = (lane_filter, lane_start, weights)
        self._dev = {}

    def lds_read_cycles(self):
        """(LDS cycles of the magnitude reads per frame under the ds_read_b128 banking model, the
        conflict-free minimum).  Equal means no bank conflicts."""
        _, lane_start, _ = self._host
        cycles = ideal = 0
        for q in range(self.n_passes):
            c, i = _pass_cycles(lane_start[q * 64:(q + 1) * 64] // 4, int(self.pass_len[q]) // 4)
            cycles += c
            ideal += i
        return cycles, ideal

    def by_filter(self, device):
        """(band_start, band_len, band_off, band_w) tensors on `device` (eligible banks only)."""
        key = "by_filter:" + str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in self._by_filter_host)
        return self._dev[key]

    def on(self, device):
        """(lane_filter, lane_start, weights) tensors on `device`."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in self._host)
        return self._dev[key]
