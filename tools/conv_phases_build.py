"""Diagnostic build: exp/libsvhip_phases.so = the current csrc/sv_conv.hip with s_memtime counters around the phases of
a pipeline step (gather issue, matrix loop, LDS hand-over, barrier).  Use with the per-workgroup trace:
    python tools/conv_phases_build.py
    SVHIP_LIB=$PWD/exp/libsvhip_phases.so SV_CONV_TRACE=gpurun_out/ph.bin python tools/conv_microbench.py --level 3 --iters 1
    python tools/conv_phases_report.py gpurun_out/ph.bin
The counters force lgkmcnt(0) at four points per step, so absolute times read ~10 % high; exp/ is git-ignored."""
import os
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(f'{root}/exp', exist_ok=True); src=open(f'{root}/markerless-robot-camera-calibration_amd/csrc/sv_conv.hip').read()
def rep(old,new):
    global src
    assert old in src, old[:60]
    src=src.replace(old,new,1)
rep('''      if (FAST || have_n) load_a(have_n ? k_n : 0, c_n, sm_n);''','''      const unsigned long long ph0 = __builtin_readcyclecounter();
      if (FAST || have_n) load_a(have_n ? k_n : 0, c_n, sm_n);
      const unsigned long long ph0b = __builtin_readcyclecounter();
      ph_issue += ph0b - ph0;''')
rep('''      ++trace_steps;
      if (!have_n) break;''','''      ++trace_steps;
      const unsigned long long ph1 = __builtin_readcyclecounter();
      ph_mfma += ph1 - ph0b;
      if (!have_n) break;''')
rep('''      store_a(As + (buf ^ 1) * (TM_ * SA), sm_n);
      k_c = k_n;''','''      store_a(As + (buf ^ 1) * (TM_ * SA), sm_n);
      const unsigned long long ph2 = __builtin_readcyclecounter();
      ph_store += ph2 - ph1;
      k_c = k_n;''')
rep('''      advance();
      __syncthreads();
      buf ^= 1;''','''      advance();
      __syncthreads();
      ph_bar += __builtin_readcyclecounter() - ph2;
      buf ^= 1;''')
rep('''  unsigned long long trace_t0 = 0;''','''  unsigned long long ph_issue = 0, ph_mfma = 0, ph_store = 0, ph_bar = 0;
  unsigned long long trace_t0 = 0;''')
rep('''    t[3] = (unsigned)trace_steps;''','''    t[0] = ph_mfma; t[1] = ph_store; t[2] = ph_issue; t[3] = ((unsigned long long)ph_bar << 24) | (unsigned)trace_steps;''')
d=f'{root}/markerless-robot-camera-calibration_amd/csrc'
open(f'{d}/sv_conv_phases_tmp.hip','w').write(src)
subprocess.check_call(f'cd {d} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I{root}/include -I. -Wno-unused-result -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -c sv_conv_phases_tmp.hip -o {root}/exp/sv_conv_phases.o && rm sv_conv_phases_tmp.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o {root}/exp/libsvhip_phases.so sv_coords.o {root}/exp/sv_conv_phases.o sv_post.o sv_dense.o sv_points.o sv_icp.o', shell=True)
print("built")
