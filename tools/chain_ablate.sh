export OPP_HIP_LIB=$PWD/onepose_plus_plus_amd/libopp_hip_tuning.so
for n in 5000 4096; do
echo "== N=$n base"; python tools/transformer_bench.py --n $n --m 64 2>&1 | grep "coarse.*True"
for a in 1 2 3; do echo "== N=$n abl $a"; OPP_CHAIN_ABL=$a python tools/transformer_bench.py --n $n --m 64 2>&1 | grep "coarse.*True"; done
for d in 2 3 6 8; do echo "== N=$n depth $d"; OPP_CHAIN_DEPTH=$d python tools/transformer_bench.py --n $n --m 64 2>&1 | grep "coarse.*True"; done
done
