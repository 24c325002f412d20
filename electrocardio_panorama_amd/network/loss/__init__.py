from .losses import losswrapper, MSELead  # noqa: F401
