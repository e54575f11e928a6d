"""IPython stand-in (only imported by the reference's plotting module)."""
