from uvad_amd.postprocess import median_filter  # noqa: F401  (GPU kernel behind uvad_median_filter)
