"""DISPATCH.md (generated on the GPU box by tools/dispatch_table.py) must stay in step with the kernels the sources define:
every ``__global__`` function is named by a row of the tables, listed as a helper, or listed as not reached by a default route."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dispatch_table_accounts_for_every_kernel():
    spec = importlib.util.spec_from_file_location("dispatch_table", os.path.join(ROOT, "tools", "dispatch_table.py"))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    assert os.path.exists(dt.DOC), "DISPATCH.md missing: gpurun -- 'python tools/dispatch_table.py'"
    errs = dt.check()
    assert not errs, "\n".join(errs)
    named, unreached = dt.parse_doc()
    # the families a default route must reach (a dispatcher change that orphans one of them should be a decision, not an accident)
    for fam in ("fit_persistent_kernel", "fit_rowlane_kernel", "fit_small_kernel", "fit_coop_kernel", "slice_pass_kernel", "fit_wide_kernel",
                "fit_wide4_kernel", "fit_wide4d_kernel", "big1_pass_kernel", "big_pass_w_kernel", "emg_chunk_kernel", "emg_wave_kernel",
                "sosfilt2_kernel", "sosfilt_chunk_kernel", "sosfilt_block_kernel"):
        assert fam in named, fam
    assert not unreached, f"kernels no default route reaches are still compiled: {sorted(unreached)} -- delete them or extend the grid"
