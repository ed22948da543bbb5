# Round 5, GPU call 3: k_sweep_xh with the books closed inside the next step's MFMA slots (compile-time chunk position): parity, timing builds, C3.
# (the script of call 4 with `for v in 0 1 2 16 64`, outputs named *_call3_*)
