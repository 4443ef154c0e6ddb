"""Do the kernels work at hidden_dim 96 / 192?  Calls the parametrised GPU tests' bodies at those widths on the library FENERF_LIB points at
(a scratch build with `case 96 / 192` in every dispatch).  usage: FENERF_LIB=/path/libfenerf_hip.so python tools/exp/h96_probe.py"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T                          # noqa: E402

cases = []
for H, grid in ((96, 5), (192, 4)):
    for prec in ("f32", "f16x3", "tape16"):
        cases.append((T.test_siren_backward_vs_autograd, ("texture", H, grid, 2, 75, prec)))
    cases.append((T.test_siren_backward_vs_autograd, ("baseline", H, 0, 1, 64, "f16x3")))
    cases.append((T.test_siren_backward_vs_autograd, ("spatial", H, 0, 2, 33, "f16x3")))
    cases.append((T.test_16bit_tape_against_the_fp32_tape, ("texture", H, grid, 2, 224)))
    for prec in ("f32", "f16x3"):
        cases.append((T.test_native_repack_is_the_torch_repack_bit_for_bit, ("texture", H, grid, prec)))
    cases.append((T.test_bf16_dump_layout_at_small_point_counts, ("texture", H, grid, 2, 224)))
    for prec in ("f16x3", "f32", "tape16", "amp"):
        cases.append((T.test_siren_backward_at_scale_vs_fp64_autograd, (prec, H, grid, 1, 40000)))
ok = bad = 0
for fn, args in cases:
    try:
        fn(*args)
        ok += 1
    except Exception as e:
        bad += 1
        print(f"FAILED {fn.__name__}{args}: {type(e).__name__}: {str(e)[:300]}")
        if os.environ.get("H96_TRACE"):
            traceback.print_exc()
print(f"h96 probe: {ok} passed, {bad} failed")
