run() { env "$@" python bench.py --steps 16 --no-cpu-baseline 2>&1 | grep -E "cumulative|metric" | sed 's/"config".*//' | cut -c1-700; }
echo "== ring0 nosync"; run AGF_WGRAD_RING=0
echo "== ring1 nosync"; run AGF_WGRAD_RING=1
echo "== ring1 sync"; run AGF_WGRAD_RING=1 AGF_BENCH_STEP_TIMES=1
echo "== ring1 nosync one-stage"; run AGF_WGRAD_RING=1 AGF_WGRAD_TWOSTAGE=0
echo "== ring1 nosync"; run AGF_WGRAD_RING=1
