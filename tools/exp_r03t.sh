#!/bin/bash
python3 examples/train_synthetic.py 2>&1 | grep -E "rl iter|validation|step 0|sup" | head -6
echo "== rank1/skip paths off"
SP_RANK1_DSP_SPLIT=0 SP_RANK1_DWC_SPLIT=0 SP_LSTM_SKIP_DPRE=0 python3 examples/train_synthetic.py 2>&1 | grep -E "rl iter|validation|step 0|sup" | head -6
