for a in "4096 64 50" "2048 16 50" "1024 1 50"; do timeout 200 python tools/predict_time.py $a 2>&1 | tail -1; done
