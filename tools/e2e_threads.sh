PYTHONPATH=. python tools/e2e_bench.py 1000000 2>&1 | grep -E "host pipeline|reader phases|end to end" | head -3
for t in 8 16 32 64; do
  echo "threads $t"
  FPLH_PARSE_THREADS=$t FPLH_TIMING=1 bin/fastplong_amd -i /tmp/e2e.fq -o /dev/null -s AAGGATTCATTCCCACGGTAACAC -e GTGTTACCGTGGGAATGAATCCTT --cut_front --cut_tail -W 5 -x -y -j /tmp/e2e.json -h /tmp/e2e.html -V 2>&1 | grep -E "host pipeline|reader phases|reports"
done
