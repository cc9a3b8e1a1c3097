"""CPU: the measurement helpers bench.py depends on (PMC summary parsing, kernel-name matching) keep working."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_pmc_traffic_lookup_is_stamped_with_the_profiled_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from committed rocprofv3 counter passes; it is reported only while the kernel sources of
    the running tree are the ones that were profiled (hash in the stamp file), and only for the profiled workload"""
    import json
    b = _bench()
    csvp, stamp = tmp_path / "t.csv", tmp_path / "s.json"
    csvp.write_text("kernel,launches,FETCH_SIZE_sum_KB,WRITE_SIZE_sum_KB,fetch_MB_per_launch_x2_corrected,write_MB_per_launch\n"
                    '"void lstm2_coop_bwd2_kernel<...>(Args)",13,1,1,1000.0,500.0\n'
                    '"void lstm2_coop_fwd_kernel<bf16, 2, 5, 64, false, true>(CoopArgs)",13,1,1,300.0,20.5\n')
    monkeypatch.setattr(b, "PMC_CSV", str(csvp))
    monkeypatch.setattr(b, "PMC_STAMP", str(stamp))
    stamp.write_text(json.dumps({"kernel_source_hash": b.kernel_source_hash(), "git_head": "abc1234"}))
    nbytes, note = b.pmc_traffic("lstm2_coop_bwd2_kernel", "C2")
    assert nbytes == 1500.0 * 1048576 and "abc1234" in note
    nbytes, _ = b.pmc_traffic(b.ROCPROF_NAME["lstm2_fwd_coop_g2[N=8224"], "C2")
    assert nbytes == 320.5 * 1048576
    assert b.pmc_traffic("lstm2_coop_bwd2_kernel", "C5")[0] is None                  # another workload: honest null
    assert b.pmc_traffic("no_such_kernel", "C2")[0] is None
    stamp.write_text(json.dumps({"kernel_source_hash": "0" * 16, "git_head": "abc1234"}))
    nbytes, note = b.pmc_traffic("lstm2_coop_bwd2_kernel", "C2")                      # kernels changed since the pass
    assert nbytes is None and "stale" in note


def test_algorithmic_work_matches_the_survey_figures():
    """bench.py's FLOP / byte formulas against SURVEY.md section 8(d)'s numbers for BASELINE C2 and C5"""
    b = _bench()
    assert abs(b.step_flops(32, 251, 5) / 1.968e13 - 1) < 5e-3
    assert abs(b.step_flops(8, 1876, 8) / 3.65e13 - 1) < 5e-3
    fam = b.hbm_family_bytes(32, 64000, 251, 5, 15_198_636, 2)
    assert abs(fam["stft"][0] / 57.7e6 - 1) < 0.02 and abs(fam["cirm_build_compress"][0] / 49.5e6 - 1) < 0.01
    assert abs(fam["cirm_decompress_apply"][0] / 57.8e6 - 1) < 0.01 and abs(fam["subband_staging_fwd"][0] / 66.6e6 - 1) < 0.01
    assert abs(fam["gs_and_loss"][0] / 230e6 - 1) < 0.02 and abs(fam["adam"][0] / 426e6 - 1) < 0.01


def test_summarize_pmc_on_a_synthetic_counter_file(tmp_path):
    hdr = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    row = "1,{d},0,1,1,1,64,1,\"{k}\",64,0,0,8,0,8,{c},{v},0,1\n"
    f, w = tmp_path / "f.csv", tmp_path / "w.csv"
    f.write_text(hdr + row.format(d=1, k="kern_a(int)", c="FETCH_SIZE", v=1024) + row.format(d=2, k="kern_a(int)", c="FETCH_SIZE", v=3072)
                 + row.format(d=3, k="kern_b()", c="FETCH_SIZE", v=10))
    w.write_text(hdr + row.format(d=1, k="kern_a(int)", c="WRITE_SIZE", v=512) + row.format(d=2, k="kern_a(int)", c="WRITE_SIZE", v=512))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), str(f), str(w)], capture_output=True,
                         text=True, check=True).stdout.strip().splitlines()
    assert out[0].startswith("kernel,launches,")
    a = out[1].split(",")
    assert a[0] == "kern_a(int)" and a[1] == "2" and a[2] == "4096" and a[3] == "1024"
    assert float(a[4]) == 2 * 4096 / 1024 / 2 and float(a[5]) == 0.5          # MB per launch: fetch doubled (gfx950), write as is
