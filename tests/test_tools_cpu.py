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


def test_pmc_traffic_lookup_matches_the_committed_summary():
    b = _bench()
    for name in ("lstm2_bwd_coop_ksplit[N=4096,T'=253,mtile=2]", "lstm2_fwd_coop_g2[N=8224,T'=253,mtile=5]",
                 "lstm2_fwd_coop_g2[N=4096,T'=253,mtile=2]"):
        nbytes, note = b.pmc_traffic(name)
        assert nbytes is not None and 1e9 < nbytes < 1e11, (name, nbytes, note)
        assert "profiles/" in note
    nbytes, note = b.pmc_traffic("lstm2_fwd_coop_g4[N=1024,T'=1878,mtile=4]")          # another configuration: honest null
    assert nbytes is None and note


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
