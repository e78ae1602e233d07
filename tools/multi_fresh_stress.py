"""Fresh-context stress of the multi-device driver (virtual ranks on one GPU): every repetition creates a NEW context and checks
its FIRST fit against the oracle — round 2's wrong results were all first fits of fresh contexts (later fits of the same problem
can read stale-but-correct data from recycled buffers).  Inputs vary per repetition for the same reason.
   python tools/multi_fresh_stress.py REPS [sk=0|1] [prio=0|1] [check=N] [comm=p2p|rccl] [hwq=16] [grids=4x2,8x1,...]
Environment is set up here (GPU_MAX_HW_QUEUES must precede HIP's initialisation).  Prints one line per failure with the
library's diagnostics and, for wrong numbers, the first wrong block column of the factor; exit status 1 on any failure."""
import os
import sys
import time
from pathlib import Path

args = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ["GPU_MAX_HW_QUEUES"] = args.get("hwq", "16")
os.environ["GPMI_COMM_PRIO"] = args.get("prio", "1")
os.environ["GPMI_MULTI_SK"] = args.get("sk", "1")
ROOT = Path(__file__).resolve().parents[1]
if args.get("comm") == "rccl":
    os.environ["GPMI_COMM"] = "rccl"
    os.environ["GPMI_RCCL_LIB"] = str(ROOT / "tests" / "rccl_mock" / "librccl_mock.so")
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

import abstractgps_jl_amd as agp  # noqa: E402
from oracle import gp_oracle as o  # noqa: E402

check = int(args.get("check", "0"))
grids = [tuple(int(v) for v in g.split("x")) for g in args.get("grids", "4x2,8x1,2x4,2x3").split(",")]
n, nb = int(args.get("n", "2049")), int(args.get("nb", "128"))
bad = tot = retries = 0
t0 = time.time()
for rep in range(reps):
    for gi, (P, Q) in enumerate(grids):
        rng = np.random.default_rng(1000 * rep + gi)
        d = int(args["dims"]) if "dims" in args else int(rng.integers(1, 4))
        X = rng.standard_normal((n, d))
        y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
        sig = float(rng.uniform(0.03, 0.3))
        depth = int(rng.integers(1, 4))
        of = o.GP(o.Kernel(o.SE, 1.0, 1.0))
        lp_ref, opost = o.logpdf_and_posterior(o.FiniteGP(of, X, sig), y)
        ctx = agp.Context(devices=[0] * (P * Q), P=P, Q=Q, nb=nb)
        ctx.set_param("lookahead_depth", depth)
        ctx.set_param("multi_timeout_s", 120)
        ctx.set_param("multi_verify", int(args.get("verify", "0")))   # raw failure rate by default; verify=1: the shipped self-check + one repetition
        if check:
            ctx.set_param("multi_check", check)
        if "copy_kernel" in args:
            ctx.set_param("copy_kernel", int(args["copy_kernel"]))
        if "dsync" in args:
            ctx.set_param("multi_debug_sync", int(args["dsync"]))
        tot += 1
        try:
            post = agp.posterior(agp.GP(agp.SqExponentialKernel(), ctx=ctx)(agp.RowVecs(X), sig), y)
            rel = abs(float(post.logpdf_value) - lp_ref) / abs(lp_ref)
            arel = float(np.linalg.norm(post.data.alpha - opost.alpha) / np.linalg.norm(opost.alpha))
            if not (rel <= 1e-10 and arel <= 1e-8):
                bad += 1
                U = post.data.C.U                               # U[j, i] = L[i, j]
                err = np.abs(U - opost.U)
                err[np.isnan(err)] = np.inf
                nbk = (n + nb - 1) // nb
                badb = [(j, i) for j in range(nbk) for i in range(j, nbk) if err[j * nb:(j + 1) * nb, i * nb:(i + 1) * nb].max() > 1e-9]  # (column block, row block), column-major
                first = badb[0] if badb else (-1, -1)
                in_first_col = [i for (j, i) in badb if j == first[0]]
                print(f"WRONG rep {rep} grid {P}x{Q} depth {depth} d {d}: logpdf rel {rel:.2e} alpha rel {arel:.2e} nan {int(np.isnan(U).sum())}; "
                      f"{len(badb)} bad L blocks of {nbk * (nbk + 1) // 2}; first bad column block {first[0]} (owner column q={first[0] % Q}), bad row blocks in it "
                      f"{in_first_col[:12]} (owner rows p={[i % P for i in in_first_col[:12]]}); max err there "
                      f"{max(err[first[0] * nb:(first[0] + 1) * nb, i * nb:(i + 1) * nb].max() for i in in_first_col) if in_first_col else 0:.2e}", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"ERROR rep {rep} grid {P}x{Q} depth {depth} d {d}: {type(e).__name__}: {str(e)[:1500]}", flush=True)
        retries += ctx.multi_stats()["retries"]
        ctx.close()
print(f"[self-check repetitions: {retries}] " if int(args.get("verify", "0")) else "", end="")
print(f"fresh-context first fits: {bad} bad of {tot} (sk={os.environ['GPMI_MULTI_SK']} prio={os.environ['GPMI_COMM_PRIO']} check={check} "
      f"comm={args.get('comm', 'copies')} hwq={os.environ['GPU_MAX_HW_QUEUES']} n={n} nb={nb}) in {time.time() - t0:.0f}s", flush=True)
sys.exit(1 if bad else 0)
