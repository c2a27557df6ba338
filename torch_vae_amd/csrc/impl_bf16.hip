// Instantiates the launch sequencing of the VAE step for bf16 storage (see vae_impl.cuh).
#include "vae_impl.cuh"
VAE_INSTANTIATE(bf16)
