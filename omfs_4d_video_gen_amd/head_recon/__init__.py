"""head_recon -- the reference's full-head reconstruction scaffolds (`02_Visual_Engine/head_recon/`):
JSON / npz bookkeeping only, no GPU work."""
