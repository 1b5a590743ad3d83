"""On-device streaming engine: the model-facing half of the reference's RTRunnerMin.step
(/root/reference/real_time_runner_minimal.py:114-167,196) for many lock-stepped IMU streams.

    eng = StreamingEngine(model, s_init)          # s_init [B,114] (q, dq) like RTRunnerMin's
    for frame in imu_frames:                       # frame [B,72]: 6 global rotations (row-major) + 6 accelerations
        out = eng.step(frame)                      # None while the 11-tap smoother primes (first 5 frames, :125-128)
        # out["s_rest"] [B,111] = s_t[3:114] (54 axis-angles, root velocity, zeros), out["c_t"] [B,20], out["y_last"]

Everything between the raw IMU frame and the fed-back history row stays in HBM: ring buffers, smoothing, root-frame
rotation, acc-sum feature, window gather, the forward pass (TIP_FWD_LAST_ROW_ONLY), output filter, SBP decode,
6D <-> axis-angle and the pose averaging (csrc/tip_stream.hip).  PyBullet FK and the SBP root-translation correction
(:169-194) remain the host's job (they never feed back into the model input).

reuse=True (include/tip_hip.h: tip_forward_reuse): a frame's model inputs never change once recorded
(real_time_runner_minimal.py:74,85,137), so its in_linear row and its layer-0 Q / K / V rows are computed ONCE, when the frame enters,
and read from a per-stream ring (40 slots x 4 KiB) in the 39 later windows it appears in: 6.8 % of a window's FLOPs, 4-5 % of a
frame at >= 1024 streams.  Exact only with the stochastic parts off — model.eval(), past_state_dropout = 0, in_dropout = 0 (the
shipped loaders run with past_state_dropout 0.8, where a frame's row differs from window to window: the engine refuses) — and then
BIT-IDENTICAL to the engine that recomputes every window on the two-window encoder (set_plan("fused2"); AUTO's own plan at 1024
streams).  reuse="auto" switches it on where that encoder is the better plan anyway (two or more windows per CU) and the model
allows it.  reset() clears the ring; a ring that does not hold the window's 40 frames yields NaN rows, never stale numbers.

use_graph=True: once the window is full (T = 40 from frame 44 on) the three calls of a frame — ingest, forward_last, consume:
about 23 kernel launches for a handful of streams, each ~8 us of host time, which is what bounds a single stream's frame rate —
are captured ONCE into a HIP graph (torch.cuda.CUDAGraph; frame / call indices come from a counter in the state buffer,
TIP_STREAM_FRAME_AUTO) and every further frame is one copy of the raw frame into a static buffer + one graph launch.  Outputs are
bit-identical to the launch-by-launch loop (tests/test_streaming_gpu.py).  The graph freezes the model's packed weights and plan:
call reset() (or build a new engine) after changing parameters.  What the graph points at is OWNED by the engine for the graph's
lifetime (its own workspace, output row, and a reference to the packed weight image), never the module's evictable per-stream
buffers.  A captured forward is outside the library's cross-stream serialisation (include/tip_hip.h): replay it on the stream
the device's other forwards use, or when none is in flight.  Every replayed frame polls the handle's hand-off word (a pinned
host word, no synchronisation): a lost hand-off of an earlier replay — its y_last was NaN and went into the history ring —
re-primes the engine (reset()), demotes the handle to the non-cooperating plans when TIP_OPT_AUTO_DEMOTE allows, and raises
TipHandoffError.  The launch-by-launch engine does the same: when the model's forward reports that it demoted itself because
of an earlier frame's loss, step() resets the engine and raises instead of consuming on top of a NaN history row.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import lib as _lib


class StreamingEngine:
    def __init__(self, model, s_init: torch.Tensor, use_graph: bool = False, reuse: bool = False):
        self.model = model
        self.use_graph = bool(use_graph)
        self.reuse = reuse          # True / False / "auto" (resolved below, once the stream count is known)
        self._graph = None
        self.lib = _lib.load()
        s_init = torch.as_tensor(s_init, dtype=torch.float32)
        if s_init.dim() == 1:
            s_init = s_init.unsqueeze(0)
        assert s_init.shape[1] == 114, "s_init is (q, dq) with 57 dofs each (constants.py:24)"
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("tip_amd.StreamingEngine runs on an MI355X: move the model to the GPU first")
        if model.training:
            import warnings
            warnings.warn("tip_amd.StreamingEngine: the model is in .train() mode — like the reference's runner without "
                          ".eval() (offline_testing_simple.py:98) every frame then draws the encoder's dropout, here through "
                          "the HIP TRAINING kernels (all rows, activation stash); call model.eval() for the inference plans")
        self.n = int(s_init.shape[0])
        nbytes = ctypes.c_size_t()
        self._check(self.lib.tip_stream_state_bytes(self.n, ctypes.byref(nbytes)))
        self.state = torch.empty(max(nbytes.value, 4), dtype=torch.uint8, device=self.device)
        self.s_init = s_init.to(self.device).contiguous()
        self.x_imu = torch.empty((self.n, 40, 90), dtype=torch.float32, device=self.device)
        self.x_s = torch.empty((self.n, 40, 131), dtype=torch.float32, device=self.device)
        self.s_rest = torch.empty((self.n, 111), dtype=torch.float32, device=self.device)
        self.c_t = torch.empty((self.n, 20), dtype=torch.float32, device=self.device)
        self.raw = torch.empty((self.n, 72), dtype=torch.float32, device=self.device)    # static input of the captured graph
        self._graph_ws = None
        self._graph_y = None
        self._ring = None
        self._ctr_ptr = None
        if self.reuse == "auto":
            # the reuse form runs on the two-window encoder (1.049 ms per round of 2 x #CUs windows, ~5.8 % less with the ring); below
            # two windows per CU the one-window kernel (0.527 ms per round of #CUs) is the better plan even without it
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            r2, rh = ((self.n + 1) // 2 + cus - 1) // cus, (self.n + cus - 1) // cus
            exact = not (model.training or model.past_state_dropout > 0.0 or model.in_dropout > 0.0)
            self.reuse = exact and self.n > cus and r2 * 1049 * 0.945 < rh * 527
        self.reuse = bool(self.reuse)
        if self.reuse:
            # a frame's in_linear row and layer-0 Q / K / V rows are kept across the 40 windows it appears in
            # (model.forward_last_reuse; exact only with the stochastic parts off — it raises otherwise)
            if model.training or model.past_state_dropout > 0.0 or model.in_dropout > 0.0:
                raise RuntimeError("tip_amd.StreamingEngine(reuse=True) needs model.eval() and a model built with past_state_dropout = 0, "
                                   "in_dropout = 0: with dropout live a frame's rows differ from window to window")
            self._ring = model.reuse_cache(self.n)
            off = ctypes.c_size_t()
            self._check(self.lib.tip_stream_frame_counter_offset(ctypes.byref(off)))
            self._ctr_ptr = self.state.data_ptr() + off.value      # the frame index the ingest kernel keeps (stream 0's block)
        self.reset()

    def _check(self, status: int):
        if status < 0:
            raise _lib.TipStatusError(status, self.lib.tip_strerror(status).decode())

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self):
        self.frame = 0
        self._graph = None
        self._graph_refs = None      # (workspace, packed weight image, y_last): what the captured kernels point at
        self._y_last = None
        if self._ring is not None:
            self.model.reuse_reset(self._ring)
        with torch.cuda.device(self.device):
            self._check(self.lib.tip_stream_reset(self.state.data_ptr(), self.s_init.data_ptr(), self.n, self._stream()))

    def _poll_handoff(self):
        """Graph mode: tip_forward's entry check never runs during a replay, so the engine reads the hand-off word itself."""
        try:
            self.model.check_handoffs(synchronize=False)
        except _lib.TipHandoffError:
            h = self.model._ensure_handle()
            if self.model._answer_handoff(h) is None:     # (launch chain instead of the one-launch form, or the plans without hand-offs)
                h.check_clear()
            self.reset()          # the NaN row of the lost frame is in the history ring: re-prime (and re-capture on the new plan)
            raise

    def _frame_auto(self):
        """ingest -> forward_last -> consume with the frame index taken from the state buffer (capturable)."""
        st = self._stream()
        # (full windows: the reuse forward reads only the newest row of x_imu / x_s — the 35-KB window gather per stream is skipped)
        ingest = self.lib.tip_stream_ingest_newest if self.reuse else self.lib.tip_stream_ingest
        self._check(ingest(self.state.data_ptr(), self.raw.data_ptr(), self.n, _lib.TIP_STREAM_FRAME_AUTO,
                           self.x_imu.data_ptr(), self.x_s.data_ptr(), st))
        if self.reuse:
            y_last = self.model.forward_last_reuse(self.x_imu, self.x_s, self._ring, 0, frame_ctr_ptr=self._ctr_ptr,
                                                   workspace=self._graph_ws, out=self._graph_y)
        else:
            y_last = self.model.forward_last(self.x_imu, self.x_s, workspace=self._graph_ws, out=self._graph_y)
        self._check(self.lib.tip_stream_consume(self.state.data_ptr(), y_last.data_ptr(), self.n, _lib.TIP_STREAM_FRAME_AUTO,
                                                self.s_rest.data_ptr(), self.c_t.data_ptr(), st))
        return y_last

    @torch.no_grad()
    def step(self, raw_imu: torch.Tensor) -> Optional[dict]:
        if self.use_graph and self.frame >= 44 and not self.model.training:
            # steady state (T = 40): one copy + one graph launch per frame
            self.raw.copy_(torch.as_tensor(raw_imu, dtype=torch.float32).reshape(self.n, 72), non_blocking=True)
            with torch.cuda.device(self.device):
                if self._graph is None:
                    # buffers the captured kernels will point at: allocated OUTSIDE the capture and held by the engine
                    if self._graph_ws is None:
                        self._graph_ws = torch.empty(self.model.workspace_bytes(self.n, 40), dtype=torch.uint8, device=self.device)
                        self._graph_y = torch.empty((self.n, self.model.size_s), dtype=torch.float32, device=self.device)
                    # packs / attaches outside the capture.  The device RNG is put back afterwards: with past_state_dropout or
                    # in_dropout live this warm-up would otherwise draw once more than the launch-by-launch loop does
                    rng = torch.cuda.get_rng_state(self.device)
                    self.model.forward_last(self.x_imu, self.x_s, workspace=self._graph_ws, out=self._graph_y)   # (never touches the ring)
                    torch.cuda.set_rng_state(rng, self.device)
                    torch.cuda.current_stream(self.device).synchronize()
                    self._poll_handoff()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._y_last = self._frame_auto()
                    self._graph = g
                    self._graph_refs = (self._graph_ws, self.model._packed_dev, self._graph_y)
                else:
                    self._poll_handoff()
                self._graph.replay()
            self.frame += 1
            return {"s_rest": self.s_rest, "c_t": self.c_t, "y_last": self._y_last, "T": 40}
        raw = torch.as_tensor(raw_imu, dtype=torch.float32).reshape(self.n, 72).to(self.device, non_blocking=True).contiguous()
        f = self.frame
        T = int(self.lib.tip_stream_window_len(f))
        with torch.cuda.device(self.device):
            # windows are written densely as [n, T, *] at the head of the preallocated buffers
            ingest = self.lib.tip_stream_ingest_newest if (self.reuse and T == 40) else self.lib.tip_stream_ingest
            self._check(ingest(self.state.data_ptr(), raw.data_ptr(), self.n, f, self.x_imu.data_ptr(), self.x_s.data_ptr(), self._stream()))
            self.frame += 1
            if T == 0:
                return None
            x_imu = self.x_imu.view(-1)[: self.n * T * 90].view(self.n, T, 90)
            x_s = self.x_s.view(-1)[: self.n * T * 131].view(self.n, T, 131)
            demotions = self.model.demotions + self.model.flow_demotions
            if self.reuse:
                try:
                    y_last = self.model.forward_last_reuse(x_imu, x_s, self._ring, f)
                except _lib.TipHandoffError:
                    # an EARLIER frame lost an inter-workgroup hand-off (its NaN row is in the history ring and in the reuse ring): same
                    # contract as the other paths — demote the handle to the plans without hand-offs when allowed, re-prime, raise
                    h = self.model._ensure_handle()
                    if self.model._answer_handoff(h) is None:
                        h.check_clear()
                    self.reset()
                    raise
            else:
                y_last = self.model.forward_last(x_imu, x_s)
            if self.model.demotions + self.model.flow_demotions != demotions:
                # tip_forward's entry check found that an EARLIER frame lost a hand-off: the model demoted itself and served this
                # call, but that frame's NaN row already went into the history ring (the prologue would scrub it to 0 for the
                # next 40 windows: finite, degraded poses).  Same contract as the graph path: re-prime and raise.
                self.reset()
                raise _lib.TipHandoffError(_lib.TIP_ERR_HANDOFF, "an earlier frame of this engine lost an inter-workgroup hand-off; "
                                           "the model now runs the non-cooperating plans and the engine was reset (re-prime it)")
            self._check(self.lib.tip_stream_consume(self.state.data_ptr(), y_last.data_ptr(), self.n, f - 5,
                                                    self.s_rest.data_ptr(), self.c_t.data_ptr(), self._stream()))
        return {"s_rest": self.s_rest, "c_t": self.c_t, "y_last": y_last, "T": T}
