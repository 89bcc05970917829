O=gpurun_out/r05_c3
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_forms" > $O/t.log 2>&1; tail -2 $O/t.log
FPL_PROF_LIB=ab_libs/prof.so PYTHONPATH=. timeout 200 python tools/prof_sections.py 1000000 8000 > $O/prof_8k.txt 2>&1; grep -A14 "k_trim_ends_batched" $O/prof_8k.txt
FPL_PROF_LIB=ab_libs/prof.so PYTHONPATH=. timeout 200 python tools/prof_sections.py 2000000 2000 > $O/prof_2k.txt 2>&1; grep -A14 "k_trim_ends_batched" $O/prof_2k.txt
