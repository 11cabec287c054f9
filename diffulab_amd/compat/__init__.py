"""stand-ins for optional third-party entry-point packages that are absent from the build / GPU images (no network)"""
