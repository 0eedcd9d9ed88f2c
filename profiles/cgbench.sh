for v in 2 4; do
  CNF_CG_NT=$v python bench.py --config cfg4 --mode grad --steps 3 --warmup 1 --preroll-seconds 0 --no-cpu-baseline 2>/dev/null | python profiles/brief.py nt$v
done
