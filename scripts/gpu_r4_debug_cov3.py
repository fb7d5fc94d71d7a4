import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_td as T
from pta_replicator_amd.engine import ReplicaEngine
import pta_replicator_amd.engine_td as et

orig = et.TimeDomainMixin.td_factorise
state = {"n": 0}
def spy(self, **kw):
    state["n"] += 1
    snap = self.d_Ltd.clone()          # stream-ordered behind the assembly, no host synchronisation
    state["snap"] = snap
    state["lay"] = [x.clone() for x in self._td_layout]
    state["K"] = (self.plan.rn_k, self.K, tuple(self.d_amp.shape), int(getattr(self, "td_cov_variant", 0)))
    state["ft"] = (bool(torch.isfinite(self.d_Ft).all().item()) if False else None)
    if os.environ.get("TWICE"):
        self.td_cov_variant = 1
        self.td_assemble()
        state["snap2"] = self.d_Ltd.clone()          # the TILE kernel, unsynchronised, same inputs
        state["inputs"] = [self._td_sigma2.clone(), (self.d_amp ** 2).clone(), self.d_Ft.clone()]
        self.td_cov_variant = 2
        self.td_assemble()
        state["snap3"] = self.d_Ltd.clone()          # the walking kernel again
    try:
        return orig(self, **kw)
    except Exception as e:
        print("td_factorise call", state["n"], "FAILED:", str(e)[-70:], flush=True)
        torch.cuda.synchronize()
        sn = state["snap"]
        v0 = sn[0:self.td_nst[0] * self.td_ld[0]].view(self.td_nst[0], self.td_ld[0])[:, :self.td_nst[0]]
        print("   snapshot: number of 7.0 in the lower triangle of J0000:", int((torch.tril(v0) == 7.0).sum().item()), "NaN:", int(torch.isnan(torch.tril(v0)).sum().item()), "of", v0.shape[0] * (v0.shape[0] + 1) // 2, flush=True)
        print("   snapshot taken (stream-ordered) between assembly and factorisation: C[0,0]", v0[0, 0].item(), "lower finite:", bool(torch.isfinite(torch.tril(v0)).all()), flush=True)
        if "snap2" in state:
            v2 = state["snap2"][0:self.td_nst[0] * self.td_ld[0]].view(self.td_nst[0], self.td_ld[0])[:, :self.td_nst[0]]
            print("   snapshot after an unsynchronised TILE-kernel assembly: C[0,0]", v2[0, 0].item(), "lower finite:", bool(torch.isfinite(torch.tril(v2)).all()), flush=True)
            v3 = state["snap3"][0:self.td_nst[0] * self.td_ld[0]].view(self.td_nst[0], self.td_ld[0])[:, :self.td_nst[0]]
            print("   snapshot after another unsynchronised WALK-kernel assembly: C[0,0]", v3[0, 0].item(), "lower finite:", bool(torch.isfinite(torch.tril(v3)).all()), flush=True)
            print("   inputs finite (sigma2, phi, Ft):", [bool(torch.isfinite(x).all()) for x in state["inputs"]], flush=True)
        print("   layout as seen on the device:", [x.cpu().tolist() for x in state["lay"]], state["K"], flush=True)
        print("   data_ptr d_Ltd", hex(self.d_Ltd.data_ptr()), "current stream", torch.cuda.current_stream().cuda_stream, flush=True)
        for a in range(self.P):
            n, ld, pos = int(self.counts[a]), self.td_ld[a], int(self.td_pos[a])
            v = self.d_Ltd[pos:pos + self.td_nst[a] * ld].view(self.td_nst[a], ld)
            lo = torch.tril(v[:, :self.td_nst[a]])
            bad = torch.nonzero(~torch.isfinite(lo))
            print("   pulsar", a, "n", n, "L[0,0]", v[0, 0].item(), "non-finite in lower:", len(bad), "first:", bad[:3].tolist(), flush=True)
        # again, synchronised
        self.td_assemble(); torch.cuda.synchronize()
        try:
            orig(self, **kw); print("   second attempt after a synchronised assembly: ok", flush=True)
        except Exception as e2:
            print("   second attempt after a synchronised assembly FAILED too:", str(e2)[-60:], flush=True)
            torch.cuda.synchronize()
            self.td_assemble(); torch.cuda.synchronize()
            v = self.d_Ltd[0:self.td_nst[0] * self.td_ld[0]].view(self.td_nst[0], self.td_ld[0])[:, :self.td_nst[0]].cpu().numpy()
            C = np.tril(v) + np.tril(v, -1).T
            print("   C[0,0]", C[0, 0], "C[-1,-1]", C[-1, -1], "finite", np.all(np.isfinite(np.tril(v))), "LAPACK:", end=" ")
            try:
                np.linalg.cholesky(C); print("ok")
            except Exception as e3:
                print("fails:", e3)
        raise
et.TimeDomainMixin.td_factorise = spy
orig_asm = et.TimeDomainMixin.td_assemble
def asm(self):
    if os.environ.get("PREFILL"):
        self.d_Ltd.fill_(7.0)
    return orig_asm(self)
et.TimeDomainMixin.td_assemble = asm
for comp in (30, 32, 29):
    try:
        T.test_td_covariance_column_walking_kernel_equals_the_tile_kernel(comp)
        print("components", comp, "ok", flush=True)
    except Exception as e:
        print("components", comp, "FAILED", str(e)[-60:], flush=True)
