cd $GRAFT_REPO_ROOT; R=$1
python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/prof_$R -o run -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-roofline --no-eager-leg > gpurun_out/${R}_prof.log 2>&1
DB=$(find gpurun_out/prof_$R -name "*.db" | head -1)
# steps in the trace: 3 eager warm-up steps (capture needs >= 3) + 2 warm replays + 10 timed replays
python tools/rocpd_stats.py $DB 15 60 > gpurun_out/${R}_bench_kernel_stats.txt
python tools/rocpd_gaps.py $DB > gpurun_out/${R}_step_gaps.txt 2>&1
# dispatches of ONE replayed step (between the last two adamw launches), by kernel: the per-step launch counts without the eager warm-up steps
( python tools/rocpd_sequence.py $DB | awk '{print $NF}' | sort | uniq -c | sort -rn; echo "total dispatches in the step: $(python tools/rocpd_sequence.py $DB | wc -l)" ) > gpurun_out/${R}_step_dispatch_hist.txt 2>&1
rm -rf gpurun_out/prof_$R
bash tools/pmc_collect.sh > gpurun_out/${R}_pmc_collect.log 2>&1
( echo "== tools/gemm_bench.py"; python tools/gemm_bench.py 2>&1 | grep -v amdgpu
  echo "== tools/attn_bench.py"; python tools/attn_bench.py 30 2>&1 | grep -v amdgpu
  echo "== tools/row_bench.py"; python tools/row_bench.py 2>&1 | grep -v amdgpu
  echo "== tools/optim_bench.py"; python tools/optim_bench.py 2>&1 | grep -v amdgpu
  echo "== tools/probes/phase_times.py"; python tools/probes/phase_times.py 2>&1 | grep "GPU\|host"
  echo "== tools/probes/l_config.py 16"; python tools/probes/l_config.py 16 2>&1 | tail -1
  echo "== tools/probes/l_config.py 64"; python tools/probes/l_config.py 64 2>&1 | tail -1
  echo "== tools/sampler_bench.py"; python tools/sampler_bench.py 2>&1 | grep sampler
  echo "== tools/sampler_bench.py --B --batch 64"; python tools/sampler_bench.py --B --batch 64 2>&1 | grep sampler
  echo "== tools/probes/fp8_bench.py"; python tools/probes/fp8_bench.py 2>&1 | grep -v amdgpu | tail -6
  echo "== tools/probes/blaslt_ref.py"; python tools/probes/blaslt_ref.py 2>&1 | grep -v amdgpu
  echo "== tools/vae_bench.py"; python tools/vae_bench.py 2>&1 | grep -v amdgpu | tail -4
  echo "== bench.py --eager"; python bench.py --no-cpu-baseline --no-roofline --eager 2>/dev/null | cut -c1-200
  echo "== bench.py (hipGraph replay, default)"; python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
  echo "== tools/probes/gemm_mainloop.sh"; bash tools/probes/gemm_mainloop.sh 2>&1
  echo "== tools/probes/guard_step.py b"; python tools/probes/guard_step.py b 2>&1 | grep guard
  echo "== tools/probes/parity_budget.py"; python tools/probes/parity_budget.py 2>&1 | grep "HIP\|reference"
) > gpurun_out/${R}_probe_outputs.txt 2>&1
# round 4: the 8-phase structure probe (32x32x16 and 16x16x32 builds, structure A/B + ablations + cycle stamps) next to the library yardstick,
# the K sweep of the shipped kernels, PMC of the attention kernels
( cd tools/probes
  [ -x gemm8p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gemm8p gemm8p.hip 2>/dev/null
  [ -x gemm8p_mf16 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DMF16=1 -o gemm8p_mf16 gemm8p.hip 2>/dev/null
  [ -x gemm8p_nostagger ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DMF16=1 -DSTAGGER=0 -o gemm8p_nostagger gemm8p.hip 2>/dev/null
  echo "== tools/probes/gemm8p 8192 (v_mfma_f32_32x32x16_bf16)"; ./gemm8p 8192 10
  echo "== tools/probes/gemm8p_mf16 8192 (v_mfma_f32_16x16x32_bf16: the shipped form)"; ./gemm8p_mf16 8192 10
  echo "== tools/probes/gemm8p_nostagger 8192 (16x16x32, both wave groups in the same phase)"; ./gemm8p_nostagger 8192 10 | head -4
  cd ../..
  echo "== tools/probes/blaslt_ref.py (hipBLASLt, same box)"; python tools/probes/blaslt_ref.py 2>&1 | grep square
  echo "== tools/gemm_bench.py (the shipped kernels through the C ABI)"; python tools/gemm_bench.py 2>&1 | grep "square\|block wgrads"
  echo "== tools/probes/gemm_ksweep.py"; python tools/probes/gemm_ksweep.py 20 2>&1 | grep -v amdgpu
) > gpurun_out/${R}_gemm8p_probe.txt 2>&1
bash tools/pmc_attn.sh > gpurun_out/${R}_pmc_attention.txt 2>&1
# round 4: the fused SwiGLU backward against its two-pass form, attention-forward workgroup geometries and the de-phased dK/dV kernel (probes build),
# config 4's end-to-end line and the VAE encode's kernel table
( echo "== tools/probes/swiglu_bwd_bench.py"; python tools/probes/swiglu_bwd_bench.py 2>&1 | grep -v amdgpu
  echo "== tools/probes/attn_fwd_geo.sh"; bash tools/probes/attn_fwd_geo.sh 2>&1 | grep -v amdgpu
  echo "== attention backward, lockstep (product) vs de-phased dK/dV kernel (probes build, MMDIT_ATTN_DKV_DP=1)"
  MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so python tools/attn_bench.py 30 2>&1 | grep "attn"
  MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so MMDIT_ATTN_DKV_DP=1 python tools/attn_bench.py 30 2>&1 | grep "attn"
) > gpurun_out/${R}_round4_ab.txt 2>&1
python tools/config4_bench.py 2>/dev/null | grep "^{" > gpurun_out/${R}_config4_line.json
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/prof_${R}_v -o run -- python3 tools/probes/vae_encode_512.py > gpurun_out/${R}_vae_prof.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_${R}_v -name "*.db" | head -1) 7 30 > gpurun_out/${R}_config4_vae_encode_kernel_stats.txt; rm -rf gpurun_out/prof_${R}_v
# MMDiT-L training step (config 4's model, batch 16) and the 28-step mxfp8 sampler (config 5): kernel tables
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/prof_${R}_l -o run -- python3 tools/probes/l_config.py 16 > gpurun_out/${R}_l_prof.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_${R}_l -name "*.db" | head -1) 8 40 > gpurun_out/${R}_l_kernel_stats.txt; rm -rf gpurun_out/prof_${R}_l
rocprofv3 --kernel-trace -d gpurun_out/prof_${R}_s -o run -- python3 tools/sampler_bench.py --batch 16 > gpurun_out/${R}_sampler_prof.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_${R}_s -name "*.db" | head -1) > gpurun_out/${R}_sampler_kernel_stats.txt; rm -rf gpurun_out/prof_${R}_s
rocprofv3 --kernel-trace -d gpurun_out/prof_${R}_s8 -o run -- python3 tools/sampler_bench.py --modes mxfp8 > gpurun_out/${R}_sampler_mxfp8_prof.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_${R}_s8 -name "*.db" | head -1) > gpurun_out/${R}_sampler_mxfp8_kernel_stats.txt; rm -rf gpurun_out/prof_${R}_s8
python -m pytest tests -m gpu -q -s 2>&1 | grep "^\[\|passed\|failed" > gpurun_out/${R}_parity_numbers.txt
# round 5: the reference's own stages (19 x 1216 at 256^2 / 512^2 / 1024^2), the fp8 / MX GEMM table, the 16x16x128 layout probe
python tools/stage_bench.py --stages 1,2,3 2>/dev/null | grep "^{" > gpurun_out/${R}_trained_stages.txt
python tools/probes/fp8_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_fp8_bench.txt
python tools/sampler_bench.py 2>&1 | grep sampler > gpurun_out/${R}_sampler_lines.txt
python tools/config4_bench.py --batch 40 2>/dev/null | grep "^{" > gpurun_out/${R}_config4_b40.json
python tools/probes/clock_under_load.py 3 2>&1 | grep -v amdgpu > gpurun_out/${R}_clock_under_load.txt
