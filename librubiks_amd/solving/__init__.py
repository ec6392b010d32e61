"""Search agents on the device engines (reference: librubiks/solving/)."""
