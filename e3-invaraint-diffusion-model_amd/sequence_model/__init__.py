"""Drop-in for the reference's sequence_model/ directory (discrete BLOSUM/uniform diffusion)."""
