# Developer tool: tests/fuzz_gpu.py over several seed offsets (each 17 622 random agent-QPs against the oracle)
for s in "$@"; do
  FUZZ_SEED0=$s python3 tests/fuzz_gpu.py > gpurun_out/fuzz_seed$s.log 2>&1
  echo "seed offset $s: $(tail -1 gpurun_out/fuzz_seed$s.log); most iterations of one agent: $(grep -o 'max iters [0-9]*' gpurun_out/fuzz_seed$s.log | sort -t' ' -k3 -n | tail -1)"
  grep -v "status mismatches 0 \[\] \[\] goal mism 0" gpurun_out/fuzz_seed$s.log | grep -v "^s64\|TOTAL\|amdgpu" | head -5
done
