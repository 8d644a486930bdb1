"""Engine behind the reference's process boundary (what `gaussian_avatars_repo/{train,render}.py` were)."""
