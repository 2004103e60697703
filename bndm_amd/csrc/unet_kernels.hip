// kernels land next commit
