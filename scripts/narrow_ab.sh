for l in dec1 dl1g logit; do for w in 0 1; do echo -n "WSC=$w "; SSC_NARROW_WSC=$w timeout 120 python scripts/conv_microbench.py $l 50; done; done
timeout 900 python -m pytest tests/test_gpu_pix2pix.py tests/test_gpu_igemm.py -m gpu -x -q 2>&1 | tail -3
