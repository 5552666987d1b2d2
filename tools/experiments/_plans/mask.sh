echo "== no mask"; timeout 120 python tools/dbg/step_time.py 65536 2>&1 | grep "N="
echo "== HSA_CU_MASK=0:0-127"; HSA_CU_MASK=0:0-127 timeout 120 python tools/dbg/step_time.py 65536 2>&1 | grep "N="
echo "== ROC_GLOBAL_CU_MASK half"; ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff timeout 120 python tools/dbg/step_time.py 65536 2>&1 | grep "N="
