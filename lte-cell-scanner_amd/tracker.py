"""LTE-Tracker's per-symbol pipeline on blocks of OFDM symbols.

The reference tracks every detected cell with one thread that consumes one OFDM symbol at a time
(src/tracker_thread.cpp:823-1068): 128 time-domain samples cut by the producer thread (src/producer_thread.cpp:
196-246) -> get_fd (:91-174) -> cell-specific reference symbols -> filter_ce (:176-201) -> frequency / timing
measurements (do_foe :203-243, do_toe_v2 :245-288) -> 2-D interpolation of the channel estimate (:383-477) -> MIB
re-decoding from the PBCH symbols of four frames (:494-529, 531-749).  Here the same work is done for a BLOCK of
symbols of MANY tracked cells at once (`Searcher.track_block`, csrc/tracker.hip); this module holds the host side:
the producer's symbol cutter, and the scalar feedback recurrences that consume the per-symbol measurements.
"""
from __future__ import annotations

import numpy as np

FS_LTE = 30720000.0


def wrap(x, lo, hi):
    """WRAP of include/macros.h:45-53: x folded into [lo, hi)."""
    return (x - lo) - (hi - lo) * np.floor((x - lo) / (hi - lo)) + lo


def n_symb_dl(cp_type: int) -> int:
    return 7 if cp_type == 1 else 6


def cut_symbols(capbuf, frame_timing: float, cp_type: int, frequency_offset: float, fc_requested: float, fc_programmed: float,
                fs_programmed: float, n_sym: int, ts_first: float = 0.0, sym_first: int = 0, pos_first: int = 0, want_state: bool = False):
    """What the producer thread queues for one tracked cell (src/producer_thread.cpp:96-131, 196-246), for a capture
    buffer whose first sample has timestamp ts_first: up to n_sym OFDM symbols starting with symbol sym_first of the stream
    (counted from slot 0 symbol 0 of its first frame), searched from sample pos_first on -- for the defaults: the symbols from
    the first frame boundary in the buffer.  Returns (td [n][128] complex128, late [n], frame_timing [n], frequency_offset
    [n]) and, with want_state, pos_next (the sample behind the last capture; sym_first + n and pos_next - o continue the
    stream on a buffer that starts o samples into this one, with timestamp wrap(ts_first + o * step, 0, 19200))."""
    cap = np.asarray(capbuf, np.complex128)
    k_factor = (fc_requested - frequency_offset) / fc_programmed
    step = (FS_LTE / 16) / (fs_programmed * k_factor)
    ts = wrap(ts_first + np.arange(cap.size) * step, 0.0, 19200.0)     # timestamp of sample n on the cell-independent 1.92 MHz time base
    td, late = [], []
    nsd = n_symb_dl(cp_type)
    target = 10.0 if cp_type == 1 else 32.0
    sym = 0
    for _ in range(int(sym_first)):                                     # (the producer's own chain; the values are integers + 10 / 32: exact)
        target = (target + (160.0 if cp_type != 1 else (138.0 if sym == 6 else 137.0))) % 19200.0
        sym = (sym + 1) % nsd
    pos = int(pos_first)
    while len(td) < n_sym:
        # first sample at or after `pos` whose timestamp is within half a sample of the target (or just past it)
        hit = -1
        limit = min(cap.size - 128, pos + 25000)
        n = pos
        while n <= limit:
            blk = ts[n:min(n + 4096, limit + 1)]
            tdiff = wrap(blk - (frame_timing + target), -9600.0, 9600.0)
            ok = np.flatnonzero((np.abs(tdiff) < 0.5) | ((tdiff > 0) & (tdiff < 3)))
            if ok.size:
                hit = n + int(ok[0])
                late.append(float(tdiff[ok[0]]))
                break
            n += blk.size
        if hit < 0:
            break
        td.append(cap[hit:hit + 128])
        pos = hit + 128
        target = (target + (160.0 if cp_type != 1 else (138.0 if sym == 6 else 137.0))) % 19200.0
        sym = (sym + 1) % nsd
    n = len(td)
    out = (np.array(td, np.complex128).reshape(n, 128), np.array(late), np.full(n, float(frame_timing)),
           np.full(n, float(frequency_offset)))
    return out + (pos,) if want_state else out


def fold_frequency_offset(f0: float, meas: np.ndarray) -> float:
    """The global frequency-offset recurrence of do_foe (src/tracker_thread.cpp:235-242) over the rows of a block's
    measurement table (columns 5, 6 = frequency_offset + residual_f, residual_f_np), in symbol order."""
    f = f0
    for row in meas[np.argsort(meas[:, 0], kind="stable")]:
        f = (f * (1 / .000001) + row[5] * (1 / row[6])) / (1 / .000001 + 1 / row[6])
    return f


def fold_frame_timing(t0: float, meas: np.ndarray) -> float:
    """The frame-timing recurrence of do_toe_v2 (src/tracker_thread.cpp:283-287); columns 7, 8 = rs_curr.frame_timing
    + delay, delay_np."""
    t = t0
    for row in meas[np.argsort(meas[:, 0], kind="stable")]:
        diff = wrap(row[7] - t, -19200.0 / 2, 19200.0 / 2)
        diff = (0 * (1 / .0001) + diff * (1 / row[8])) / (1 / .0001 + 1 / row[8])
        t = (t + diff) - 19200.0 * np.floor((t + diff) / 19200.0)
    return t


def fold_ac_fd(ac0: np.ndarray, ac_fd_rows: np.ndarray, meas_rows: np.ndarray) -> np.ndarray:
    """tracked_cell.ac_fd's update (src/tracker_thread.cpp:335-338) over the rows of one port of a block, in row order:
    ac_fd_np = (np^2/sp^2 + 2 np/sp) / (12, 11, ..., 1); ac = (ac / 1e-5 + ac_fd / ac_fd_np) / (1 / 1e-5 + 1 / ac_fd_np).
    ac_fd_rows [n][12] from lcs_track_stats, meas_rows [n][9] (columns 1 = np, 4 = sp)."""
    ac = np.array(ac0, np.complex128)
    div = np.arange(12.0, 0.0, -1.0)
    for row, m in zip(ac_fd_rows, meas_rows):
        w = 1.0 / ((m[1] * m[1] / (m[4] * m[4]) + 2 * m[1] / m[4]) / div)
        ac = (ac * (1 / .00001) + row * w) / (1 / .00001 + w)
    return ac


def fold_ac_td(ac0: np.ndarray, ac_td_rows: np.ndarray) -> np.ndarray:
    """tracked_cell.ac_td's update (src/tracker_thread.cpp:367-368) over the rows that have a full history (not NaN)."""
    ac = np.array(ac0, np.complex128)
    for row in ac_td_rows:
        if np.isnan(row[0]):
            continue
        ac = (ac * (1 / .00001) + row * 1 / 1) / (1 / .00001 + 1)
    return ac


def fold_sync_power(av, sync_rows: np.ndarray):
    """The sync_{tp,sp,np,np_blank}_av recurrences of do_pss_sss_sigpower_ce (src/tracker_thread.cpp:808-818): the first
    PSS/SSS pair initialises the averages (av = None or NaN), every later one moves them by 0.001."""
    av = None if av is None or np.isnan(np.asarray(av, np.float64)[1]) else np.array(av, np.float64)
    for row in sync_rows:
        av = np.array(row, np.float64) if av is None else 0.999 * av + .001 * row
    return av


def mib_lock_walk(mib_ok, failures: float = 0.0, synchronized: bool = False, drop_threshold: float = 400.0):      # CELL_DROP_THRESHOLD, include/constants.h:35
    """do_mib_decode's fifo walk (src/tracker_thread.cpp:552-745) over a block in which every frame offset has been
    tried in parallel: an attempt is made whenever 16 PBCH symbols are queued; success or a synchronised failure
    consumes four frames, an unsynchronised failure one.  `mib_ok` is a row of track_block's 'mib_ok' codes as they
    come (bit 0 = CRC, bit 1 = bandwidth / PHICH fields equal the tracked cell's, -1 = not attempted): only code 3 is a
    lock (:689-694 needs both) and the walk ends at the first offset that was never attempted (the reference would be
    waiting for more symbols there).  A boolean sequence (True = locked) is accepted too.
    Returns (failures, synchronized, attempts made, dropped)."""
    a = np.asarray(mib_ok)
    if a.dtype == np.bool_:
        codes = np.where(a, 3, 0)
    elif np.issubdtype(a.dtype, np.integer):
        codes = a
        if codes.size and (codes.min() < -1 or codes.max() > 3):
            raise ValueError("mib_lock_walk: integer input must be track_block's mib_ok codes (-1, 0..3); pass booleans for plain lock flags")
    else:
        raise ValueError("mib_lock_walk takes track_block's integer mib_ok codes or booleans")
    o, attempts = 0, 0
    n = len(codes)
    while o < n and codes[o] != -1:
        attempts += 1
        if codes[o] == 3:
            synchronized, failures, o = True, 0.0, o + 4
        elif synchronized:
            failures, o = failures + 1.0, o + 4
        else:
            failures, o = failures + 0.25, o + 1
        if failures >= drop_threshold:
            return failures, synchronized, attempts, True
    return failures, synchronized, attempts, False


class DeviceTracker:
    """LTE-Tracker's producer thread + tracker threads as ONE data path with the samples on the device (round 6).

    The reference (src/producer_thread.cpp, src/tracker_thread.cpp) moves every sample through a fifo, cuts each tracked cell's
    OFDM symbols out on the host and queues them to that cell's thread.  Here the dongle's bytes go up once per buffer (0.3 MB per
    80 ms) and stay in HBM: `push` appends a buffer to the stream, cuts the symbols of all tracked cells there (Searcher.track_cut,
    continuing from the previous buffer: timestamp of the first kept sample, next symbol, next search position per cell -- the tail
    of the previous buffer is kept so that a capture its end cut off is whole), runs the tracker block on them where they lie
    (Searcher.track_stream_block on the device pointer) and folds the measurements into the two slow loops the reference closes
    around the producer: the GLOBAL frequency offset (one crystal: do_foe's recurrence over every cell's rows, src/tracker_thread.cpp:
    235-242; it sets the time base's step, src/producer_thread.cpp:99, 127) and every cell's frame timing (do_toe_v2, :283-287).

    cells: searcher records (n_id_1/2, cp_type, n_ports, n_rb_dl, PHICH fields); frame_timing [n_cells] on the 1.92 MHz time base
    (frame_start * (FS_LTE/16) / (fs_programmed * k_factor), src/searcher_thread.cpp:224); frequency_offset: the global one.
    feedback=False holds both (a recorded buffer replayed open loop).  Needs torch for the device buffers (plumbing)."""

    KEEP = 512      # samples kept behind the last complete capture at least (a capture is 128 samples, a window opens < 4 samples wide)

    def __init__(self, searcher, cells, frame_timing, frequency_offset, fc_requested, fc_programmed, fs_programmed, device=0, feedback=True):
        import torch
        self._torch, self.S, self.cells = torch, searcher, list(cells)
        self.dev = torch.device("cuda", device)
        self.fc, self.fcp, self.fsp, self.feedback = float(fc_requested), float(fc_programmed), float(fs_programmed), bool(feedback)
        self.frame_timing = np.array(frame_timing, np.float64).reshape(len(self.cells))
        self.frequency_offset = float(frequency_offset)
        self.cp = np.array([int(c.cp_type) for c in self.cells], np.int32)
        self.ts_first = 0.0                                        # timestamp of the first sample of the kept buffer
        self.sym_next = np.zeros(len(self.cells), np.int64)        # next symbol of each cell (counted from slot 0 symbol 0 of the stream's first frame)
        self.pos_next = np.zeros(len(self.cells), np.int64)        # ... and where its search starts in the kept buffer
        self._buf = torch.empty(0, dtype=torch.uint8, device=self.dev)
        self._td = None
        self.mib_codes = [[] for _ in self.cells]
        self.lock = [(0.0, False, 0, False) for _ in self.cells]
        self.symbols_done = 0
        searcher.track_stream_reset()

    def _step(self):
        return (FS_LTE / 16) / (self.fsp * ((self.fc - self.frequency_offset) / self.fcp))

    def push(self, iq_u8, want_ce=False):
        """iq_u8: the next bytes of the stream (2 per sample), a host array or a torch uint8 tensor already on the device.
        Returns None when no cell has a complete symbol yet, else the dict of Searcher.track_stream_block for the symbols cut from
        the stream so far (+ 'n_sym', 'late'); the loops (frequency_offset, frame_timing, lock) are updated on the object."""
        torch = self._torch
        new = iq_u8 if isinstance(iq_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(iq_u8, np.uint8))
        self._buf = torch.cat([self._buf, new.to(self.dev).reshape(-1)])
        n_cap = self._buf.numel() // 2
        if n_cap < 128:
            return None
        C_ = len(self.cells)
        fo = np.full(C_, self.frequency_offset)
        probe = n_cap // 137 + 2
        if self._td is None or self._td.shape[1] < probe:
            self._td = torch.empty((C_, probe, 128), dtype=torch.complex128, device=self.dev)
        cut = lambda n: self.S.track_cut(self._buf.data_ptr(), 1, n_cap, self.cp, self.frame_timing, fo, self.fc, self.fcp, self.fsp, n,
                                         self._td.data_ptr(), ts_first=self.ts_first, sym_first=self.sym_next, pos_first=self.pos_next, want_state=True)
        _, n_cut, _ = cut(probe)
        n = int(n_cut.min())                                       # the block moves all cells by the same number of symbols
        out = None
        pos_next = self.pos_next
        if n > 0:
            late, _, pos_next = cut(n)                             # [cell][n][128] contiguous; pos_next = behind symbol n - 1 of every cell
            out = self.S.track_stream_block(self.cells, None, np.repeat(fo[:, None], n, 1), np.repeat(self.frame_timing[:, None], n, 1), late[:, :n],
                                            self.fc, self.fcp, self.fsp, want_syms=False, want_ce=want_ce, td_device_ptr=self._td.data_ptr())
            out["n_sym"], out["late"] = n, late[:, :n]
            self.sym_next = self.sym_next + n
            self.symbols_done += n
            for i in range(C_):
                self.mib_codes[i] += [int(v) for v in out["mib_ok"][i, :out["n_mib"][i]]]
                self.lock[i] = mib_lock_walk(np.array(self.mib_codes[i], np.int32)) if self.mib_codes[i] else self.lock[i]
            if self.feedback:
                for i in range(C_):                                # port 0's filtered reference symbols drive the loops
                    m = out["meas"][i, 0, :out["n_meas"][i, 0]]
                    self.frequency_offset = fold_frequency_offset(self.frequency_offset, m)
                    self.frame_timing[i] = fold_frame_timing(self.frame_timing[i], m)
        # drop what every cell is done with, keep the tail; the time base moved on with the step in force while the samples were stamped
        adv = int(max(0, min(int(pos_next.min()), n_cap - self.KEEP)))
        step = (FS_LTE / 16) / (self.fsp * ((self.fc - fo[0]) / self.fcp))
        self.ts_first = float(wrap(self.ts_first + adv * step, 0.0, 19200.0))
        self.pos_next = pos_next - adv
        self._buf = self._buf[2 * adv:].clone()
        return out
