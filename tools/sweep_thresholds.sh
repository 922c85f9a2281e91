for ts in 16 24 32 40 48 56; do for tl in 8 16 24 32 48; do
 echo -n "TS=$ts TL=$tl "; RTIOW_TS=$ts RTIOW_TL=$tl timeout -k 10 120 python -u bench.py --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['mean_radiance'],5))"
done; done
