// Instantiates the launch sequencing of the VAE step for float storage (see vae_impl.cuh).
#include "vae_impl.cuh"
VAE_INSTANTIATE(float)
