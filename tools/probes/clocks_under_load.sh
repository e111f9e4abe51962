# Run ON the GPU box: GPU clock and socket power while the headline bench leg runs (why the conv kernels sit at half the
# nominal MFMA peak: DESIGN.md 3.1).   gpurun -- 'bash tools/probes/clocks_under_load.sh'
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower 2>&1 | grep -i "power (w)"
python bench.py --train-steps 0 --no-cpu-baseline --steps 3000 --warmup 20 --no-extra-legs --no-kernel-timing > /dev/null 2>&1 &
pid=$!
for i in $(seq 1 120); do
  p=$(rocm-smi --showpower 2>/dev/null | grep -i "power (w)" | grep -o "[0-9.]*$")
  if [ "${p%.*}" -gt 500 ] 2>/dev/null; then break; fi
  sleep 1
done
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|socket" | tr '\n' ' '; echo
  sleep 1
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
