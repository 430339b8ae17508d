set -x
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_driver.py -m gpu -q -x ) 2>&1 | tail -3
for T in 32 64; do
( time timeout 500 python profiles/scripts/e2e_driver.py 30e6 2e8 $T ) > $O/e2e_$T.log 2>&1; grep -E "mtb_classify:" $O/e2e_$T.log | cut -c1-400
done
