"""TEST INFRASTRUCTURE for tests/test_bench_spawn.py: stands in for BatchedMPC with the CPU build of the solver header
(tests/host_twin) so that bench.py's multi-process plumbing -- the self-spawn of the ranks, the torch.distributed
rendezvous, the packed gather, max-over-ranks timing, rank 0's JSON line -- can be exercised in a container without a
GPU.  bench.py only imports this module when it is called with --stub; the line it then prints is marked as a stub."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Stats:
    pass


def make(kind, pkg):
    assert kind == "host_twin", kind
    d = os.path.join(ROOT, "tests", "host_twin")
    subprocess.check_call(["make", "-s", "-C", d])
    twin = C.CDLL(os.path.join(d, "libhost_twin.so"))

    class BatchedMPC:
        def __init__(self, params, max_batch):
            self.params = params.copy()
            self.max_batch = max_batch
            self._last = None

        def solve_torch(self, state, coeffs, yaw_lo, yaw_hi, weights=None, want_traj=False, outputs=None, stream=None):
            import torch
            assert not state.is_cuda
            f32 = self.params.precision == pkg.PRECISION_F32
            B = state.shape[1]
            out, traj, status, iters = outputs["out"], outputs.get("traj"), outputs["status"], outputs["iters"]
            fn = twin.mpc_host_twin_solve_f32 if f32 else twin.mpc_host_twin_solve
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            ld = out.stride(0)
            assert state.stride(0) == B
            # outputs are views into the packed buffer (leading dimension = its row length): solve into dense temporaries
            o = torch.empty((9, B), dtype=out.dtype); tr = torch.empty((2 * self.params.N, B), dtype=out.dtype) if traj is not None else None
            st = torch.empty(B, dtype=torch.int32); it = torch.empty(B, dtype=torch.int32)
            rc = fn(C.byref(self.params), C.c_int64(B), C.c_int64(B), p(state), p(coeffs), p(yaw_lo), p(yaw_hi), p(weights), p(o), p(tr), p(st), p(it))
            assert rc == 0
            out.copy_(o); status.copy_(st); iters.copy_(it)
            if traj is not None:
                traj.copy_(tr)
            self._last = (st.numpy().copy(), it.numpy().copy())
            return outputs

        def stats(self):
            s = _Stats()
            st, it = self._last
            s.batch = len(st); s.iter_sum = int(it.sum()); s.iter_max = int(it.max()); s.n_success = int((st == 0).sum()); s.kernel_ms = 0.0
            return s

        def close(self):
            pass

    class Stub:
        pass
    Stub.BatchedMPC = BatchedMPC
    return Stub
