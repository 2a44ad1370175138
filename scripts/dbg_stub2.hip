namespace dmp { int g_exact_fp32 = 0; }
