"""Model front-ends: thin handles on the native graph runner (``ssd_net_*``) with the reference's ``get_model`` / decoder interface."""
