"""Drop-in alias package for the reference `methods` package."""
