for P in 3 4 6 9; do for C in cfg4 nv20; do
echo "PER_CU=$P $C: $(CNF_LG_WGRAD_PER_CU=$P timeout 300 python bench.py --config $C --mode grad --steps 5 --warmup 2 --no-cpu-baseline --preroll-seconds 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done; done
