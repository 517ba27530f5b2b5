#!/bin/bash
echo "== default"; python tools/reprocheck.py 3 2>/dev/null
echo "== UBR_DEFER_REDUCE=0"; UBR_DEFER_REDUCE=0 python tools/reprocheck.py 3 2>/dev/null
echo "== UBR_WGRAD_PC=0"; UBR_WGRAD_PC=0 python tools/reprocheck.py 3 2>/dev/null
echo "== UBR_CONV_FAST_EPI=0"; UBR_CONV_FAST_EPI=0 python tools/reprocheck.py 3 2>/dev/null
echo "== UBR_WGRAD_STREAM=0"; UBR_WGRAD_STREAM=0 python tools/reprocheck.py 3 2>/dev/null
