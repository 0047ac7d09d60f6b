"""BatchedMPC: thin Python handle over the C ABI (include/mpc_amd.h).

Mirrors, for a batch, what ``MPC::solve(state, target_velocity, x_traj, y_traj)``
returns in the reference (src/control/MPC.cpp:183-325): a 9-vector per instance
``{x1,y1,psi1,v1,cte1,epsi1,delta0,a0,cost}`` plus the optional N-point (x,y)
trajectory, and a status per instance (the reference only prints it).
torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import MpcBatchStats, MpcParams, check, library


class BatchedMPC:
    def __init__(self, params: MpcParams, max_batch: int, device: int = -1):
        self.params = params.copy()
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        check(library().mpc_create(C.byref(self.params), int(device), self.max_batch, C.byref(self._h)), "mpc_create")

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            library().mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def N(self):
        return self.params.N

    @property
    def f32(self):
        """True for a handle created with params.precision = MPC_PRECISION_F32 (float32 tensors in and out)."""
        return self.params.precision == _abi.PRECISION_F32

    def _dtype(self):
        import torch
        return torch.float32 if self.f32 else torch.float64

    def set_params(self, params: MpcParams):
        check(library().mpc_set_params(self._h, C.byref(params)), "mpc_set_params")
        self.params = params.copy()

    # -- device path (torch tensors resident in HBM) -------------------------
    def alloc_outputs(self, B, device, want_traj=False):
        import torch
        dt = self._dtype()
        out = {
            "out": torch.empty((_abi.NOUT, B), dtype=dt, device=device),
            "status": torch.empty((B,), dtype=torch.int32, device=device),
            "iters": torch.empty((B,), dtype=torch.int32, device=device),
            "traj": torch.empty((2 * self.N, B), dtype=dt, device=device) if want_traj else None,
        }
        return out

    def solve_torch(self, state, coeffs, yaw_lo, yaw_hi, weights=None, want_traj=False, outputs=None, stream=None):
        """state [6,B], coeffs [5,B], yaw_lo/hi [B], weights [12,B] or None: CUDA tensors, float64 (float32 for a
        handle created with precision F32).  Asynchronous on ``stream`` (default: torch's current stream).  Returns the
        dict of output tensors."""
        import torch
        B = state.shape[1]
        dt = self._dtype()
        for name, t, rows in (("state", state, 6), ("coeffs", coeffs, 5)):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous() or t.shape != (rows, B):
                raise ValueError("%s must be a contiguous %s CUDA tensor of shape (%d, B)" % (name, dt, rows))
        for name, t in (("yaw_lo", yaw_lo), ("yaw_hi", yaw_hi)):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous() or t.shape != (B,):
                raise ValueError("%s must be a contiguous %s CUDA tensor of shape (B,)" % (name, dt))
        if weights is not None and (weights.dtype != dt or not weights.is_cuda or
                                    not weights.is_contiguous() or weights.shape != (_abi.NW, B)):
            raise ValueError("weights must be a contiguous %s CUDA tensor of shape (12, B)" % dt)
        if outputs is None:
            outputs = self.alloc_outputs(B, state.device, want_traj)
        s = stream if stream is not None else torch.cuda.current_stream(state.device)
        traj = outputs.get("traj")
        fn = library().mpc_solve_batch_device_f32 if self.f32 else library().mpc_solve_batch_device
        check(fn(self._h, B, B, state.data_ptr(), coeffs.data_ptr(), yaw_lo.data_ptr(), yaw_hi.data_ptr(),
                 weights.data_ptr() if weights is not None else None, outputs["out"].data_ptr(),
                 traj.data_ptr() if traj is not None else None, outputs["status"].data_ptr(),
                 outputs["iters"].data_ptr(), C.c_void_p(s.cuda_stream)), "mpc_solve_batch_device")
        return outputs

    def run_torch(self, pose, ptsx, ptsy, want_traj=False, want_pre=False, stream=None):
        """MPC::run() for a batch on the device (src/control/MPC.cpp:327-382): pose [6,B] = x,y,psi,v,steering,
        acceleration; ptsx/ptsy [npts,B] global waypoints, overwritten with the vehicle-frame waypoints as the
        reference does.  Returns out8 [8,B] = {x1,y1,psi1,v1,steer in [-1,1],accel,cte1,epsi1} etc."""
        import torch
        B = pose.shape[1]
        npts = ptsx.shape[0]
        for name, t in (("pose", pose), ("ptsx", ptsx), ("ptsy", ptsy)):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.shape[1] != B:
                raise ValueError("%s must be a contiguous float64 CUDA tensor [rows, B]" % name)
        dev = pose.device
        res = {"out8": torch.empty((8, B), dtype=torch.float64, device=dev),
               "status": torch.empty((B,), dtype=torch.int32, device=dev),
               "iters": torch.empty((B,), dtype=torch.int32, device=dev),
               "traj": torch.empty((2 * self.N, B), dtype=torch.float64, device=dev) if want_traj else None,
               "pre": torch.empty((15, B), dtype=torch.float64, device=dev) if want_pre else None}
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        check(library().mpc_run_batch_device(
            self._h, B, B, int(npts), pose.data_ptr(), ptsx.data_ptr(), ptsy.data_ptr(), res["out8"].data_ptr(),
            res["traj"].data_ptr() if want_traj else None, res["status"].data_ptr(), res["iters"].data_ptr(),
            res["pre"].data_ptr() if want_pre else None, C.c_void_p(s.cuda_stream)), "mpc_run_batch_device")
        return res

    def run_numpy(self, pose, ptsx, ptsy, want_traj=False):
        """MPC::run() for host arrays (mpc_run_batch_host; B = 1 is what include/mpc_drop_in.hpp's MPC::run calls): pose [6,B],
        ptsx / ptsy [npts,B] global waypoints.  Returns out8, status, iters, pre [15,B], the vehicle-frame waypoints and traj."""
        import numpy as np
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        pose, px, py = f(pose), f(ptsx).copy(), f(ptsy).copy()
        B, npts = pose.shape[1], px.shape[0]
        out8 = np.empty((8, B)); pre = np.empty((15, B)); status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        traj = np.empty((2 * self.N, B)) if want_traj else None
        p = lambda a: a.ctypes.data if a is not None else None
        check(library().mpc_run_batch_host(self._h, B, B, int(npts), p(pose), p(px), p(py), p(out8), p(traj), p(status), p(iters), p(pre)),
              "mpc_run_batch_host")
        return {"out8": out8, "status": status, "iters": iters, "pre": pre, "ptsx": px, "ptsy": py, "traj": traj}

    def telemetry_torch(self, tel, ptsx, ptsy, extra_latency=0.0, want_out8=False, stream=None):
        """The telemetry handler around run() (src/mpc_main.cpp:126-174) for a batch: tel [6,B] = x, y, psi,
        speed [mph], steering_angle (simulator sign), previous throttle.  Returns cmd [2,B] = (steering_angle,
        throttle) of the reply, status, and optionally run()'s 8-vector."""
        import torch
        B = tel.shape[1]
        npts = ptsx.shape[0]
        for name, t in (("tel", tel), ("ptsx", ptsx), ("ptsy", ptsy)):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.shape[1] != B:
                raise ValueError("%s must be a contiguous float64 CUDA tensor [rows, B]" % name)
        dev = tel.device
        res = {"cmd": torch.empty((2, B), dtype=torch.float64, device=dev),
               "status": torch.empty((B,), dtype=torch.int32, device=dev),
               "out8": torch.empty((8, B), dtype=torch.float64, device=dev) if want_out8 else None}
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        check(library().mpc_telemetry_batch_device(
            self._h, B, B, int(npts), tel.data_ptr(), float(extra_latency), ptsx.data_ptr(), ptsy.data_ptr(),
            res["cmd"].data_ptr(), res["out8"].data_ptr() if want_out8 else None, res["status"].data_ptr(),
            C.c_void_p(s.cuda_stream)), "mpc_telemetry_batch_device")
        return res

    def rollout_torch(self, state, coeffs, yaw_lo, yaw_hi, steps, weights=None, want_hist=True, stream=None):
        """Closed loop of src/test.cpp:79-111 for a batch: `steps` cold-started solves, each fed with the previous
        step-1 state.  `state` [6,B] is advanced in place.  Returns hist [steps,9,B], worst status, summed iters."""
        import torch
        B = state.shape[1]
        dev = state.device
        for name, t, shape in (("state", state, (6, B)), ("coeffs", coeffs, (5, B)), ("yaw_lo", yaw_lo, (B,)), ("yaw_hi", yaw_hi, (B,))):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != shape:
                raise ValueError("%s must be a contiguous float64 CUDA tensor of shape %s" % (name, shape))
        res = {"hist": torch.empty((steps, _abi.NOUT, B), dtype=torch.float64, device=dev) if want_hist else None,
               "status": torch.empty((B,), dtype=torch.int32, device=dev),
               "iters": torch.empty((B,), dtype=torch.int32, device=dev)}
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        check(library().mpc_rollout_batch_device(
            self._h, B, B, int(steps), state.data_ptr(), coeffs.data_ptr(), yaw_lo.data_ptr(), yaw_hi.data_ptr(),
            weights.data_ptr() if weights is not None else None, res["hist"].data_ptr() if want_hist else None,
            res["status"].data_ptr(), res["iters"].data_ptr(), C.c_void_p(s.cuda_stream)), "mpc_rollout_batch_device")
        return res

    # -- host path (numpy arrays; copies through PCIe) ------------------------
    def solve_numpy(self, state, coeffs, yaw_lo, yaw_hi, weights=None, want_traj=False):
        """Host arrays through mpc_solve_batch_host (float64), or mpc_solve_batch_host_f32 for an MPC_PRECISION_F32 handle."""
        dt = np.float32 if self.f32 else np.float64
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dt))
        state, coeffs, yaw_lo, yaw_hi = f(state), f(coeffs), f(yaw_lo), f(yaw_hi)
        B = state.shape[1]
        assert state.shape == (6, B) and coeffs.shape == (5, B) and yaw_lo.shape == (B,) and yaw_hi.shape == (B,)
        if weights is not None:
            weights = f(weights)
            assert weights.shape == (_abi.NW, B)
        out = np.empty((_abi.NOUT, B), dtype=dt); status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        traj = np.empty((2 * self.N, B), dtype=dt) if want_traj else None
        p = lambda a: a.ctypes.data if a is not None else None
        fn = library().mpc_solve_batch_host_f32 if self.f32 else library().mpc_solve_batch_host
        check(fn(self._h, B, B, p(state), p(coeffs), p(yaw_lo), p(yaw_hi), p(weights), p(out), p(traj), p(status), p(iters)), "mpc_solve_batch_host")
        return {"out": out, "status": status, "iters": iters, "traj": traj}

    # -- deferred tails (MpcParams.tail_cut > 0, include/mpc_amd.h) -------------
    def last_batch_id(self):
        return int(library().mpc_last_batch_id(self._h))

    def tail_wait(self, batch_id=0):
        """Block until batch `batch_id` is final (0: every batch issued so far)."""
        check(library().mpc_tail_wait(self._h, int(batch_id)), "mpc_tail_wait")

    def tail_poll(self, batch_id):
        """True if batch `batch_id` is final (one non-blocking turn of the pump that starts tail slices)."""
        r = library().mpc_tail_poll(self._h, int(batch_id))
        if r < 0:
            check(r, "mpc_tail_poll")
        return bool(r)

    def tail_stream_wait(self, batch_id, stream):
        check(library().mpc_tail_stream_wait(self._h, int(batch_id), C.c_void_p(stream.cuda_stream)), "mpc_tail_stream_wait")

    def tail_flush(self):
        check(library().mpc_tail_flush(self._h), "mpc_tail_flush")

    def tail_pending(self, batch_id):
        n = C.c_int64(0)
        check(library().mpc_tail_pending(self._h, int(batch_id), C.byref(n)), "mpc_tail_pending")
        return int(n.value)

    def tail_info(self):
        a = (C.c_int64 * 13)()
        check(library().mpc_tail_info(self._h, a), "mpc_tail_info")
        return {"batches_deferred": int(a[0]), "tail_launches": int(a[1]), "ring": int(a[2]), "capacity_per_batch": int(a[3]),
                "waves_per_tail_launch": int(a[4]), "tail_stream_high_priority": bool(a[5]), "tail_streams": int(a[6]),
                "batches_not_deferred_survivors_full": int(a[7]), "passes_per_slice": int(a[8]), "survivors": int(a[9]),
                "tail_cut_in_use": int(a[10]), "deferred_share": (int(a[11]) / 65536.0 if a[11] >= 0 else None), "queue_overflows": int(a[12])}

    def synchronize(self):
        check(library().mpc_synchronize(self._h), "mpc_synchronize")

    def stats(self) -> MpcBatchStats:
        st = MpcBatchStats()
        check(library().mpc_get_stats(self._h, C.byref(st)), "mpc_get_stats")
        return st
