"""Host-side mirrors of the reference's ``utils`` package for the parts built on the device."""
