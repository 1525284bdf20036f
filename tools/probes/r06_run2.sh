#!/bin/bash
# Round 6, GPU call 2: the whole -m gpu suite after the knob clean-up and with the new shape-trained tolerance tests; the slot-refill price model.
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_b_gputests.txt
echo "gpu tests rc ${PIPESTATUS[0]}"
tail -5 gpurun_out/r06_b_gputests.txt
timeout 900 python tools/probes/slot_refill_model.py head trained driver1000 soak3000 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_b_slot_refill_model.txt
echo "model rc $?"
cat gpurun_out/r06_b_slot_refill_model.txt
