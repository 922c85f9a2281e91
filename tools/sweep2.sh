for ts in 56 64; do for tl in 4 8 16 24 32; do
 echo -n "w3 TS=$ts TL=$tl "; RTIOW_HIP_LIB=$PWD/variants/librtiow_hip_w3.so RTIOW_TS=$ts RTIOW_TL=$tl timeout -k 10 120 python -u bench.py --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['mean_radiance'],5))"
done; done
