"""Placeholder import target until the EDM wrapper lands (filled in below in this round)."""


class ElucidatedImagen:   # replaced by the real class once built
    pass
