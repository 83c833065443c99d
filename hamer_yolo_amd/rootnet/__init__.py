"""RootNet root-depth regressor (reference: rootnet/Model_RGB.py, rootnet/preprocessing.py) -- SURVEY.md 8(f) rank 1."""
