"""one-off stress run of the randomized differential tests with many more seeds than the suite uses
    python tools/stress_fuzz.py [n] [first seed]     (GPU box; exits non-zero on the first failure)"""
import inspect, os, sys, traceback
import pytest
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
start = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
only = os.environ.get("FUZZ_ONLY")
fns = [T.test_random_sequences_vs_oracle, T.test_random_fused_sequences_vs_oracle, T.test_random_nd_sequences_vs_oracle,
       T.test_random_jacobians_vs_oracle, T.test_packed_kernel_is_bit_identical, T.test_packed_jacobians_vs_oracle,
       T.test_random_trains_vs_oracle, T.test_random_repetition_trains_vs_oracle,
       T.test_random_vectorised_nd_sequences_vs_oracle, T.test_random_single_variable_jacobians, T.test_random_fused_jacobians_vs_oracle,
       T.test_random_repetition_trains_with_derivatives, T.test_random_long_trains_vs_oracle]
bad = 0
for fn in fns:
    if only and only not in fn.__name__:
        continue
    raw = getattr(fn, "__wrapped__", fn)
    for seed in range(start, start + n):
        try:
            if "monkeypatch" in inspect.signature(raw).parameters:
                with pytest.MonkeyPatch.context() as mp:
                    raw(seed, mp)
            else:
                raw(seed)
        except Exception:
            bad += 1
            print(f"FAIL {fn.__name__} seed={seed}")
            traceback.print_exc(limit=3)
            if bad >= 5:
                sys.exit(1)
    print(fn.__name__, "ok", n, "seeds", flush=True)
sys.exit(1 if bad else 0)
