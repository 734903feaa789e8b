for i in 1 3 7; do python tools/gemm_bench.py 2>/dev/null | tail -18 | sed -n "$((i+1))p" | sed -E "s/ \| t128:[^|]*//; s/ \| t13[01]:[^|]*//g; s/e=[0-9e+.-]*//g"; done
