#!/bin/bash
# robustness check on the GPU box: write a large gzip output (parallel members), read it back as input, compare the headline counts
PYTHONPATH=. python tools/e2e_bench.py 100000 2>&1 | grep -E "wrote|end to end" | head -2
A="-s AAGGATTCATTCCCACGGTAACAC -e GTGTTACCGTGGGAATGAATCCTT --cut_front --cut_tail -W 5 -x -y"
T0=$(date +%s%N); bin/fastplong_amd -i /tmp/e2e.fq -o /tmp/rt.fq.gz -A -Q -L -j /tmp/rt0.json -h /tmp/rt0.html >/dev/null 2>&1; echo "write .gz: $(( ($(date +%s%N) - T0) / 1000000 )) ms"
ls -la /tmp/e2e.fq /tmp/rt.fq.gz | awk '{print $5, $9}'
T0=$(date +%s%N); bin/fastplong_amd -i /tmp/rt.fq.gz -o /tmp/rt2.fq $A -j /tmp/rt_gz.json -h /tmp/rt_gz.html >/dev/null 2>&1; echo "gz input: $(( ($(date +%s%N) - T0) / 1000000 )) ms"
T0=$(date +%s%N); bin/fastplong_amd -i /tmp/e2e.fq -o /tmp/rt3.fq $A -j /tmp/rt_plain.json -h /tmp/rt_plain.html >/dev/null 2>&1; echo "plain input: $(( ($(date +%s%N) - T0) / 1000000 )) ms"
cmp /tmp/rt2.fq /tmp/rt3.fq && echo "outputs identical"
grep -v '"command"' /tmp/rt_gz.json | md5sum; grep -v '"command"' /tmp/rt_plain.json | md5sum
