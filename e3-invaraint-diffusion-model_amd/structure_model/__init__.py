"""Drop-in for the reference's structure_model/ directory (angle-space DDPM denoiser)."""
