// Instantiates the launch sequencing of the VAE step for f16 storage (see vae_impl.cuh).
#include "vae_impl.cuh"
VAE_INSTANTIATE(f16)
