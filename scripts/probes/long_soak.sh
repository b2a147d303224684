set -u
O=gpurun_out/r6_long; mkdir -p $O
timeout -k 10 700 python3 scripts/soak.py --seconds 400 > $O/soak_plain.txt 2>&1; echo "soak plain rc $?"; tail -3 $O/soak_plain.txt
timeout -k 10 700 python3 scripts/soak.py --seconds 400 --holes 0.11 > $O/soak_compacted.txt 2>&1; echo "soak compacted rc $?"; tail -3 $O/soak_compacted.txt
