"""Empty stand-in for wandb (fixture generation only)."""
