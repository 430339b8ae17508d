# A/B runs of bench.py with environment knobs / library variants; prints stage times from stderr
for v in "MTB_LIB=metabuli_amd/csrc/libmtb.so" "MTB_LIB=metabuli_amd/csrc/libmtb_vA.so" "MTB_LIB=metabuli_amd/csrc/libmtb.so" "MTB_LIB=metabuli_amd/csrc/libmtb_vA.so"; do
  env $v timeout 300 python bench.py --steps 2 --warmup 1 --no-parity > gpurun_out/var.json 2> gpurun_out/var.err
  echo "$v: $(grep 'stage ms' gpurun_out/var.err | cut -c1-160)"
done
