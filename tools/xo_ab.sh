set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crossover or step or envelope" > gpurun_out/xo_parity.log 2>&1
tail -3 gpurun_out/xo_parity.log
: > gpurun_out/xo_ab.log
for rep in 1 2 3; do
 for v in 0 1; do
  for u in 4 8; do
   echo "VARIANT=$v U=$u rep=$rep" >> gpurun_out/xo_ab.log
   GNX_XO_VARIANT=$v GNX_XO_UNROLL=$u timeout -k 10 200 python tools/kbench.py --genomes --steps 10 2>/dev/null | grep -E "crossover|ms/step=" >> gpurun_out/xo_ab.log
  done
 done
done
cat gpurun_out/xo_ab.log
