"""bench.py's host-side helpers that need no GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_pmc_counter_csv_is_reduced_to_per_launch_means(tmp_path):
    """rocprofv3 --pmc writes one row per launch and counter; the bench wants the mean per kernel BASE name (all batch
    widths of a template together), and only of the counter asked for."""
    d = tmp_path / "host" / "123"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,"
            "Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,"
            "Start_Timestamp,End_Timestamp"]
    def row(name, ctr, val):
        return '1,1,1,1,1,1,1024,7,"%s",256,0,0,64,0,32,%s,%s,0,1' % (name, ctr, val)
    rows += [row("void bioen::k_strip_adj<8, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "FETCH_SIZE", 4000000.0),
             row("void bioen::k_strip_adj<1, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "FETCH_SIZE", 4200000.0),
             row("void bioen::k_strip_adj<8, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "SQ_WAVES", 7.0),
             row("void bioen::k_strip_fwd<8, true>(bioen::StripArgs, bioen::Vec8)", "FETCH_SIZE", 1.0),
             row("bioen::k_gram(bioen::GramArgs, int, bioen::Xch)", "FETCH_SIZE", 5.0)]
    (d / "pmc_counter_collection.csv").write_text("\n".join(rows) + "\n")
    means = bench.pmc_kernel_means(str(tmp_path), "FETCH_SIZE")
    assert means["k_strip_adj"] == (4100000.0, 2)
    assert means["k_strip_fwd"] == (1.0, 1) and means["k_gram"] == (5.0, 1)
    assert bench.pmc_kernel_means(str(tmp_path), "WRITE_SIZE") == {}


def test_survey_inputs_follow_the_recipe_stream():
    """ONE default_rng(12345) stream -- YTrue, the matrix row by row, then the targets."""
    M, N = 5, 40
    y, Y = bench.survey_inputs(M, N, seed=12345)
    rng = np.random.default_rng(12345)
    YTrue = rng.uniform(1, 10, M)
    first_row = rng.normal(YTrue[0], 0.5 * YTrue[0], N) / (0.1 * YTrue[0])
    assert y.shape == (M, N) and Y.shape == (M,) and np.array_equal(y[0], first_row)
    assert abs(y[3].mean() - 10.0) < 5.0                      # ytilde_i ~ N(10, 5): mean YTrue_i / (0.1 YTrue_i)


def test_profiler_detection_reads_the_environment(monkeypatch):
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES", raising=False)
    monkeypatch.setenv("LD_PRELOAD", "/some/guard.so")
    monkeypatch.delenv("HSA_TOOLS_LIB", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/some/guard.so:/opt/rocm-7.2.0/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
