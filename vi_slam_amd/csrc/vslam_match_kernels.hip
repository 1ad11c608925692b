/* placeholder until matcher kernels land */
