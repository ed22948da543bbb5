timeout 60 tools/ubench/mfma_f64_rate.bin
